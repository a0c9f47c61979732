#!/bin/bash
# The headline with the tree kernels in the tree against an earlier engine.hip (ataxxzero_amd/csrc/_ab_engine_prev.hip: e.g. `git show <rev>:ataxxzero_amd/csrc/engine.hip`,
# plus a shim for any symbol it lacks), same box, same call, three interleaved rounds.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
V=$(python3 -c "
import os
from ataxxzero_amd import build
print(build.build_variant('round4engine', [], replace={'engine.hip': os.path.abspath('ataxxzero_amd/csrc/_ab_engine_prev.hip')}))") || exit 1
for round in 1 2 3; do
  for lib in product round4engine; do
    if [ $lib = product ]; then unset AZH_LIB; else export AZH_LIB=$V; fi
    timeout -k 10 200 python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-gemm-ceiling --no-target-leg | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round %-12s two half-batches: %.3f M node-evals/s, tower %.4f over the chip, tree phase per half %.4f ms, per launch %.4f ms' % ('$lib', d['value']/1e6, d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['roofline']['avg_launch_ms']))" || exit 3
  done
done
