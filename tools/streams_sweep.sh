#!/bin/bash
# node-evals/s for several (games, half-batches in flight) pairs; run on the GPU box from the repo root
for cfg in "4096 1" "4096 2" "4608 3" "8192 2" "12288 3" "16384 1" "16384 2"; do
  set -- $cfg
  timeout -k 10 280 python bench.py --games $1 --streams $2 --no-cpu-baseline --no-target-leg --no-gemm-ceiling | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('games %5d streams %d: %.3f M node-evals/s  %.3f ms/iteration  tower %.3f ms (%.1f%%)  tree %.3f ms' % ($1, $2, d['value']/1e6, d['ms_per_iteration'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration']))" || exit 1
done
