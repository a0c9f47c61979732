#!/bin/bash
# Two games per wave (csrc/tree_halfwave.h) against one wave per game at 16384 games, and the register budget of the
# half-wave kernel: a copy of the library per AZH_TREE_H_OCC value (waves per SIMD the compiler must leave room for;
# 0 = no bound) and bench.py's 16384-game workload with each, on ONE device in ONE call; AZH_TREE_LANES=64 first (the
# one-wave-per-game kernel in rounds).  usage (GPU box, repo root): bash tools/tree_lanes_sweep.sh 0 6 7 8
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
GAMES=${GAMES:-16384}
line() {
  python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1, %5d games: %.3f M node-evals/s  %.3f ms/iteration  tower %.3f ms (%.1f%%)  tree phase %.4f ms  tree roofline %.3f  parked %.3f' % ($GAMES, d['value']/1e6, d['ms_per_iteration'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['tree_roofline']['frac'], d['counters']['parked']/max(1,d['counters']['parked']+d['counters']['steps'])))"
}
ARGS="--games $GAMES --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-target-leg --no-gemm-ceiling ${EXTRA_ARGS}"
AZH_TREE_LANES=64 python3 bench.py $ARGS | line "64 lanes per game" || exit 1
for occ in "$@"; do
  lib=$(python3 -c "from ataxxzero_amd import build; print(build.build_variant('treehocc$occ', ['-DAZH_TREE_H_OCC=$occ']))") || exit 2
  AZH_LIB=$lib AZH_TREE_LANES=32 python3 bench.py $ARGS | line "32 lanes per game, occupancy bound $occ" || exit 1
done
