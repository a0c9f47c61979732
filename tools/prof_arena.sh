#!/bin/bash
# rocprofv3 kernel trace of the config-5 arena leg (bench.py --legs config5_arena): how long the tower, the tree launch and the
# re-roots take as the match thins out, and what an iteration's kernel boundaries cost.  GPU box, repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${OUT:-$R/gpurun_out/prof_arena}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-ceiling --legs config5_arena > $OUT/trace.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# the arena leg = everything from the first k_tower2_pair on
first = next(i for i, r in enumerate(rows) if "k_tower2_pair" in r[0])
rows = rows[first:]
def pct(v, p): v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))] / 1e3
for name in ("k_tower2_pair", "k_tree", "k_advance_list"):
    d = [e - s for n, s, e in rows if name in n]
    if not d: continue
    n = len(d)
    print("%-16s %6d launches: mean %.1f us  p10 %.1f  p50 %.1f  p90 %.1f | first tenth of the match mean %.1f us, last half mean %.1f us" % (
        name, n, sum(d) / n / 1e3, pct(d, .1), pct(d, .5), pct(d, .9), sum(d[:n // 10]) / (n // 10) / 1e3, sum(d[n // 2:]) / (n - n // 2) / 1e3))
# iteration period and the idle time between kernels on the main stream, over the last half (the thin tail)
tw = [(s, e) for n, s, e in rows if "k_tower2_pair" in n]
half = tw[len(tw) // 2:]
per = [b[0] - a[0] for a, b in zip(half, half[1:])]
per = [p for p in per if p < 2e6]   # (drop the host round trips between runs of 25 iterations)
print("tower start -> next tower start over the last half: mean %.1f us  p50 %.1f us (tower p50 %.1f us of it)" % (
    sum(per) / len(per) / 1e3, pct(per, .5), pct([e - s for s, e in half], .5)))
PY
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
