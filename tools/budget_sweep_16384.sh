#!/bin/bash
# The select level budget at BASELINE's >= 10k-games point (16384 games): node-evals/s, tree phase and its roofline fraction
# per budget; run on the GPU box from the repo root.  (tools/budget_sweep.sh is the same at the headline's 4096 games.)
for b in ${BUDGETS:-16 24 32 40 48 64}; do
  timeout -k 10 280 python3 bench.py --games 16384 --steps 6 --warmup 2 --select-budget $b --no-cpu-baseline --no-target-leg --no-gemm-ceiling | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['counters']
print('budget %3d: %.3f M node-evals/s  %.3f ms/iteration  tower %.3f ms (%.1f%%, %.0f evals/launch)  tree %.4f ms  tree roofline %.3f  parked %.3f  plies/s %.0f' % ($b, d['value']/1e6, d['ms_per_iteration'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac'], d['roofline']['evals_per_launch'], d['tree_roofline']['tree_phase_ms_per_iteration'], d['tree_roofline']['frac'], c['parked']/max(1,c['parked']+c['steps']), d['plies_per_s']))" || exit 1
done
