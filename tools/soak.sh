#!/bin/bash
# Full-scale soak of the drop-in generator: 12x128 random-init net, 400 visits, default buffer
# (4096 games in two half-batches in flight, the generator's default; EXTRA="--streams 1" for one batch) for $1 seconds; prints rate lines and game stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
SECS=${1:-120}
OUT=$R/gpurun_out/soak
mkdir -p $OUT
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from ataxxzero_amd import model
conv, bn = model.random_init(12, 128, seed=1)
model.save_model("$OUT/model-001.npy", conv, bn)
PY
rm -f $OUT/model-001-0.json
python3 $R/accelerated_generate_games.py --network $OUT/model-001.npy --output-games $OUT/model-001-0.json --visits ${VISITS:-400} --max-seconds $SECS ${EXTRA} > $OUT/log.txt 2>&1
echo "exit $?"
grep "Rate:" $OUT/log.txt | tail -5
python3 - <<PY
import json
n=0; pl=[]; res={1:0,2:0}
for line in open("$OUT/model-001-0.json"):
    if line.strip():
        e=json.loads(line); n+=1; pl.append(len(e["moves"])); res[e["result"]]+=1
print("games", n, "mean plies", sum(pl)/max(n,1), "min/max", (min(pl), max(pl)) if pl else None, "results", res)
PY
ls -la $OUT/model-001-0.json; rm -f $OUT/model-001-0.json $OUT/model-001.npy
rocm-smi --showmeminfo vram 2>/dev/null | head -8
