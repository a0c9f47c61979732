#!/bin/bash
# (Rounds 3-5's loop only — since round 6 the queued moves ride in the tower launch; export AZH_REROOT_SIDE_STREAM=1 to trace the old loop.)
# Does a re-root launch (k_advance_list, side stream) hold up the tree launch that waits for it?  It starts beside a tower
# and the next k_tree of its engine waits for both: the hold-up is how far the re-root launch outlasts that tower.
# rocprofv3 kernel trace of bench.py with ARGS (default: configs[3] = 8x128, f16, 800 sims/move as the headline, two
# half-batches in flight); engines are told apart by their HIP streams.  GPU box, repo root.
#   ARGS="--visits 200" NAME=config2 bash tools/reroot_holdup.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=${NAME:-config4}
OUT=${OUT:-$R/gpurun_out/reroot_holdup_$NAME}
ARGS=${ARGS:---visits 800 --blocks 8 --dtype f16}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-target-leg --no-gemm-ceiling > $OUT/trace.log 2>&1 || exit 1
python3 - "$OUT" "$ARGS" > $OUT/summary.txt <<'PY'
import bisect, collections, csv, glob, os, sys
out, args = sys.argv[1], sys.argv[2]
f = max(glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Stream_Id"]))
rows.sort()
print("bench.py %s --steps 3 --warmup 1 (rocprofv3 --kernel-trace; the last 3 x 250 iterations of every engine = the timed region)" % args)
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r[3]].append(r)
mains = [s for s, v in by_stream.items() if sum(1 for r in v if r[2].startswith("k_tower")) > 700]
sides = [s for s, v in by_stream.items() if sum(1 for r in v if r[2] == "k_advance_list") > 700]
mains.sort(key=lambda s: by_stream[s][0][0])
sides.sort(key=lambda s: next(r[0] for r in by_stream[s] if r[2] == "k_advance_list"))
def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))] / 1e3
tot_iters = tot_late = 0
tot_hold = 0.0
for i, (ms, ss) in enumerate(zip(mains, sides)):
    tw = [r for r in by_stream[ms] if r[2].startswith("k_tower")][-750:]
    tr = [r for r in by_stream[ms] if r[2] == "k_tree"][-750:]
    t_lo = tw[0][0]
    adv = [r for r in by_stream[ss] if r[2] == "k_advance_list" and r[0] >= t_lo]
    starts = [t[0] for t in tw]
    late, slack = [], []
    for s0, e0, _, _ in adv:
        k = bisect.bisect_right(starts, s0 + 30000) - 1      # the tower launched beside it (their starts are microseconds apart)
        if k >= 0 and e0 - s0 < 5000000 and abs(tw[k][0] - s0) < 200000:
            # (a re-root launch is matched with the tower that started within 0.2 ms of it; the pairs across a host round
            # trip — the drain between two steps — are left out)
            late.append(max(0, e0 - tw[k][1]))
            slack.append(tw[k][1] - e0)
    n_late = sum(1 for x in late if x > 0)
    tot_iters += len(late); tot_late += n_late; tot_hold += sum(late)
    print("engine %d (streams %s / %s): tower p50 %.1f us mean %.1f | k_tree p50 %.1f us | k_advance_list p50 %.1f p90 %.1f max %.1f us" % (
        i, ms, ss, pct([e - s for s, e, _, _ in tw], .5), sum(e - s for s, e, _, _ in tw) / len(tw) / 1e3,
        pct([e - s for s, e, _, _ in tr], .5), pct([e - s for s, e, _, _ in adv], .5), pct([e - s for s, e, _, _ in adv], .9),
        pct([e - s for s, e, _, _ in adv], 1.0)))
    print("  re-root launches that end after the tower they run under: %d of %d (%.1f %%); they outlast it by %.1f us on average "
          "when they do (max %.1f us) = %.2f us per iteration over all iterations; slack when they do not: p10 %.0f us p50 %.0f us" % (
              n_late, len(late), 100.0 * n_late / max(1, len(late)), sum(late) / max(1, n_late) / 1e3, max(late) / 1e3 if late else 0.0,
              sum(late) / max(1, len(late)) / 1e3, pct([x for x in slack if x > 0] or [0], .1), pct([x for x in slack if x > 0] or [0], .5)))
print("all engines: %d of %d iterations (%.1f %%) have a re-root launch that outlasts its tower; hold-up %.2f us per iteration" % (
    tot_late, tot_iters, 100.0 * tot_late / max(1, tot_iters), tot_hold / max(1, tot_iters) / 1e3))
PY
st=$(ls $OUT/trace/*/*_kernel_stats.csv | head -1)
echo >> $OUT/summary.txt; echo "rocprofv3 --stats (whole process: set-up, warm-up and the timed region):" >> $OUT/summary.txt
head -8 "$st" >> $OUT/summary.txt
tail -3 $OUT/trace.log | cut -c1-1500 >> $OUT/summary.txt
rm -rf $OUT/trace
cat $OUT/summary.txt
