#!/bin/bash
# A/B of a tree-kernel build variant on ONE box in ONE call (boxes differ by several percent): the product library against
# a copy compiled with extra flags — in-kernel stamps of the tree launch at 4096 and 16384 games, then bench.py's one-batch
# headline and 16384-game leg, interleaved, two rounds.
#   bash tools/tree_variant_ab.sh fastscore -DAZH_FAST_SCORE=1    # the level's chain without IEEE division / square root:
#        a MEASUREMENT build (not bit-exact with the oracle), upper bound of what precomputing q = W/n and r = cP/(1+n) at
#        backup time could buy for a level (round-4 review, item 4)
#   bash tools/tree_variant_ab.sh sqrtlate -DAZH_SQRT_EARLY=0     # sqrt(1 + N) behind the arrival of the level's records
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
name=$1; shift
# ENGINE_SRC=<file>: the variant's engine.hip comes from that file (an earlier version of the tree kernels)
V=$(python3 -c "
import os, sys
from ataxxzero_amd import build
src = os.environ.get('ENGINE_SRC')
print(build.build_variant(sys.argv[1], sys.argv[2:], replace={'engine.hip': os.path.abspath(src)} if src else None))" "$name" "$@") || exit 1
echo "variant '$name' ($*): $V"
for games in 4096 16384; do
  for lib in product $name; do
    if [ $lib = product ]; then unset AZH_LIB; else export AZH_LIB=$V; fi
    echo "== tree stamps, $lib, $games games"
    timeout -k 10 300 python3 tools/tree_stamps.py --games $games --samples 8 || exit 2
  done
done
for round in 1 2; do
  for lib in product $name; do
    if [ $lib = product ]; then unset AZH_LIB; else export AZH_LIB=$V; fi
    timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-gemm-ceiling --legs target_10k_games |
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['target_10k_games']
print('round $round %-10s 4096 games, one batch: %.3f M node-evals/s, tree phase %.4f ms (frac %.3f), tower %.3f ms | 16384 games: %.3f M, tree %.4f ms (frac %.3f)' % (
  '$lib', d['value']/1e6, d['tree_roofline']['tree_phase_ms_per_iteration'], d['tree_roofline']['frac'], d['roofline']['avg_launch_ms'],
  t['node_evals_per_s']/1e6, t['tree_ms_per_iteration'], t['tree_roofline_frac']))" || exit 3
  done
done
