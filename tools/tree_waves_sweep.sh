#!/bin/bash
# Games per workgroup of the fused tree kernel (AZH_TREE_WAVES): builds a copy of the library per value and runs bench.py's
# (--streams 1: the kernels alone on the chip, one batch; bench.py's default is two half-batches in flight)
# 4096-game and 16384-game workloads with each, on ONE device in ONE call.  usage (GPU box, repo root): bash tools/tree_waves_sweep.sh 1 2 4 8
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for w in "$@"; do
  lib=$(python3 -c "from ataxxzero_amd import build; print(build.build_variant('treewaves$w', ['-DAZH_TREE_WAVES=$w']))") || exit 2
  for g in 4096 16384; do
    steps=8; [ $g = 16384 ] && steps=3
    AZH_LIB=$lib python3 bench.py --streams 1 --games $g --steps $steps --warmup 2 --no-cpu-baseline --no-target-leg --no-gemm-ceiling | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('games/workgroup $w, %5d games: %.3f M node-evals/s  %.3f ms/iteration  tower %.3f ms (%.1f%%)  tree phase %.4f ms  tree roofline %.3f' % ($g, d['value']/1e6, d['ms_per_iteration'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['tree_roofline']['frac']))" || exit 1
  done
done
