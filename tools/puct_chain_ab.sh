#!/bin/bash
# A/B on ONE box in ONE call: the PUCT level's dependent chain with IEEE divisions / square root (the product, bit-exact with
# the oracle) against single approximate instructions (-DAZH_FAST_SCORE=1: a measurement build, never shipped) — an upper
# bound of what precomputing q = W/n and r = cP/(1+n) at backup time could buy for a level (round-4 review, item 4).
# In-kernel stamps of the tree launch at 4096 and 16384 games, then the bench's one-batch and 16384-game legs, interleaved.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
V=$(python3 -c "from ataxxzero_amd import build; print(build.build_variant('fastscore', ['-DAZH_FAST_SCORE=1']))") || exit 1
echo "variant library: $V"
for games in 4096 16384; do
  for lib in product fast_score; do
    if [ $lib = fast_score ]; then export AZH_LIB=$V; else unset AZH_LIB; fi
    echo "== tree stamps, $lib, $games games"
    timeout -k 10 300 python3 tools/tree_stamps.py --games $games --samples 8 || exit 2
  done
done
for round in 1 2; do
  for lib in product fast_score; do
    if [ $lib = fast_score ]; then export AZH_LIB=$V; else unset AZH_LIB; fi
    timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-gemm-ceiling --legs target_10k_games |
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['target_10k_games']
print('round $round %-10s 4096 games, one batch: %.3f M node-evals/s, tree phase %.4f ms (frac %.3f), tower %.3f ms | 16384 games: %.3f M, tree %.4f ms (frac %.3f)' % (
  '$lib', d['value']/1e6, d['tree_roofline']['tree_phase_ms_per_iteration'], d['tree_roofline']['frac'], d['roofline']['avg_launch_ms'],
  t['node_evals_per_s']/1e6, t['tree_ms_per_iteration'], t['tree_roofline_frac']))" || exit 3
  done
done
