#!/bin/bash
# Timeline of one steady-state iteration from rocprofv3's kernel trace of bench.py: durations and the gaps between
# tower -> k_tree -> next tower, and of the side stream's re-root launch (run on the GPU box from the repo root).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/gap_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-target-leg --no-gemm-ceiling > $OUT/trace.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = max(glob.glob(os.path.join(sys.argv[1], "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]) for r in csv.DictReader(open(f))]
rows.sort()
main = [r for r in rows if r[2] in ("k_tower2", "k_tree")]
tail = main[-2 * 300:]
import collections
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for a, b in zip(tail, tail[1:]):
    durs[a[2]].append(a[1] - a[0])
    gaps[a[2] + " -> " + b[2]].append(b[0] - a[1])
for k, v in durs.items():
    v.sort(); print("duration %-10s p50 %.1f us  mean %.1f us" % (k, v[len(v)//2]/1e3, sum(v)/len(v)/1e3))
for k, v in gaps.items():
    v.sort(); print("gap %-24s p50 %.1f us  mean %.1f us" % (k, v[len(v)//2]/1e3, sum(v)/len(v)/1e3))
adv = [r for r in rows if r[2] == "k_advance_list"][-300:]
v = sorted(e - s for s, e, _ in adv); print("k_advance_list (side stream) p50 %.1f us  p90 %.1f us  max %.1f us" % (v[len(v)//2]/1e3, v[9*len(v)//10]/1e3, v[-1]/1e3))
# does a re-root launch ever hold up the tree launch that waits for it?  It runs under a tower; the tree launch behind that
# tower can start once both have ended: the hold-up is how far the re-root launch outlasts the tower it started under
import bisect
towers = [r for r in rows if r[2] == "k_tower2"]
starts = [t[0] for t in towers]
late = []
for s0, e0, _ in adv:
    i = bisect.bisect_right(starts, s0 + 20000) - 1     # the tower launched beside it (their starts are microseconds apart)
    if i >= 0:
        late.append(max(0, e0 - towers[i][1]))
n_late = sum(1 for x in late if x > 0)
print("re-root launches that end after the tower they run under: %d of %d; the tree launch behind is held up by %.2f us on average over all iterations (max %.1f us)" % (
    n_late, len(late), sum(late) / max(1, len(late)) / 1e3, max(late) / 1e3 if late else 0.0))
PY
rm -rf $OUT/trace
