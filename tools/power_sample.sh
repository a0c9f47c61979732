#!/bin/bash
# Samples rocm-smi power / clocks while the net-only bench loops (run on the GPU box from the repo root).
R=${GRAFT_REPO_ROOT:-$PWD}
python3 - <<PY &
import sys; sys.path.insert(0, "$R")
from ataxxzero_amd import link, model
conv, bn = model.random_init(12, 128, seed=1)
net = link.Net(conv, bn)
for _ in range(12):
    net.bench(16384, iters=200, dtype=link.DTYPES["bf16"])
PY
PID=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo "--"
  sleep 1.5
done
wait $PID
