#!/bin/bash
# BASELINE.json configs beside the headline (run on the GPU box from the repo root): C2 (200 sims), C4 (8x128 f16,
# 800 sims), C5 (1000-game arena of two 12x128 nets, 100 visits per move).
R=${GRAFT_REPO_ROOT:-$PWD}
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: %.3f M node-evals/s  %.3f ms/iteration  tower %.1f%% of peak  tree %.3f ms  plies/s %.0f  games/s %.1f' % (d['value']/1e6, d['ms_per_iteration'], 100*d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['plies_per_s'], d['games_per_s']))"; }
timeout -k 10 280 python3 bench.py --visits 200 --no-cpu-baseline --no-target-leg --no-gemm-ceiling | show "C2  4096 games, 200 sims, 12x128 bf16" || exit 1
timeout -k 10 400 python3 bench.py --visits 800 --blocks 8 --dtype f16 --no-cpu-baseline --no-target-leg --no-gemm-ceiling | show "C4  4096 games, 800 sims,  8x128 f16 " || exit 2
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from ataxxzero_amd import model
for seed, name in ((1, "a"), (2, "b")):
    conv, bn = model.random_init(12, 128, seed=seed)
    model.save_model("/tmp/arena-%s.npy" % name, conv, bn)
PY
t0=$(date +%s.%N)
timeout -k 10 900 python3 uai_ringmaster.py --engine "python uai_interface.py --network-path /tmp/arena-a.npy --visits ${ARENA_VISITS:-100}" --engine "python uai_interface.py --network-path /tmp/arena-b.npy --visits ${ARENA_VISITS:-100}" --game-count 1000 --pgn-out /tmp/arena.pgn > /tmp/arena.log 2>&1 || { tail -5 /tmp/arena.log; exit 3; }
t1=$(date +%s.%N)
echo "C5  1000-game arena, two 12x128 nets, ${ARENA_VISITS:-100} visits/move: $(python3 -c "print('%.1f' % ($t1-$t0))") s wall  ($(grep -c FinalScore /tmp/arena.pgn) PGN games)  last: $(grep Wins: /tmp/arena.log | tail -1)"
