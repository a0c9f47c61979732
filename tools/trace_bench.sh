#!/bin/bash
# kernel trace of bench.py: per-kernel average / late-run average / max durations
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# a step is 250 search iterations: keep the trace to a few thousand launches, and keep bench.py's child processes (the
# vendor-GEMM leg) and extra legs out of it — a child inherits the profiler and writes a trace of its own
ARGS=${ARGS:---steps 6 --warmup 2 --no-target-leg --no-gemm-ceiling}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline ${ARGS} > $OUT/log.txt 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
import os
f=max(glob.glob("gpurun_out/prof_trace/*/*_kernel_trace.csv"), key=os.path.getsize)   # the bench process, not a helper's
rows=list(csv.DictReader(open(f)))
by=collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,d in by.items():
    d2=d[-len(d)//4:]
    print("%-44s n=%-6d avg %8.1f us   last-quarter avg %8.1f us   max %8.1f us"%(k[:44], len(d), sum(d)/len(d)/1e3, sum(d2)/len(d2)/1e3, max(d)/1e3))
PY
grep "^{" $OUT/log.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f ms/step %.3f'%(d['value'], d['ms_per_step']))"
find $OUT -name "*_kernel_trace.csv" -delete
