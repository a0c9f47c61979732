#!/bin/bash
# node-evals/s against the select level budget; run on the GPU box from the repo root
for b in ${BUDGETS:-0 16 24 32 48 64}; do
  timeout -k 10 280 python bench.py --select-budget $b --no-cpu-baseline --no-target-leg --no-gemm-ceiling | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('budget %3d: %.3f M node-evals/s  %.3f ms/iteration  tower %.3f ms (%.1f%%, %.0f evals/launch)  tree %.3f ms  plies/s %.0f' % ($b, d['value']/1e6, d['ms_per_iteration'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac'], d['roofline']['evals_per_launch'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['plies_per_s']))" || exit 1
done
