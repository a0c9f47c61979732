#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_tree
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/tree_phases.py > $OUT/trace.log 2>&1 || exit 1
f=$(ls -t $OUT/trace/*/*_kernel_stats.csv | head -1)
cat $f
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
f = max(glob.glob(os.path.join(sys.argv[1], "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v[-300:])
    print("%-40s last300: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (k[-40:], v[len(v)//10]/1e3, v[len(v)//2]/1e3, v[9*len(v)//10]/1e3, v[-1]/1e3))
PY
find $OUT -name "*_kernel_trace.csv" -delete
