#!/bin/bash
# A/B of two builds of the library on ONE box in ONE call (boxes differ by several percent): the product library against a
# prebuilt variant (AZH_LIB), bench.py's headline + the legs named, interleaved rounds.
#   bash tools/lib_ab.sh NAME /path/to/variant.so [rounds] [legs]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
name=$1; V=$(realpath $2); rounds=${3:-2}; legs=${4:-one_batch,config2,config4}
for round in $(seq 1 $rounds); do
  for lib in product $name; do
    if [ $lib = product ]; then unset AZH_LIB; else export AZH_LIB=$V; fi
    timeout -k 10 400 python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-gemm-ceiling --legs $legs | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s='round $round %-10s headline %.3f M node-evals/s (tower %.4f over the chip, tree phase per half %.4f ms)' % ('$lib', d['value']/1e6, d['roofline']['frac'], d['tree_roofline']['tree_phase_ms_per_iteration'])
for k in '$legs'.split(','):
    if k in d and 'node_evals_per_s' in d[k]:
        s += ' | %s %.3f M (tower %.4f, tree %.4f ms)' % (k, d[k]['node_evals_per_s']/1e6, d[k]['tower_frac_of_peak'], d[k]['tree_ms_per_iteration'])
    elif k in d and 'wall_s' in d[k]:
        s += ' | %s %.2f s' % (k, d[k]['wall_s'])
print(s)" || exit 3
  done
done
