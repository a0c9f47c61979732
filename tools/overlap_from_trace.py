#!/usr/bin/env python3
"""How many tower launches are in flight at a time?  From a rocprofv3 kernel trace of bench.py (tools/prof_bench.sh writes
one): the share of the timed region with 0, 1, 2 ... launches of k_tower running.  rocprofv3's tracing holds the two engines'
dispatches apart to a degree (the traced run is slower than the plain one), so this is a LOWER bound of the overlap.
usage: python tools/overlap_from_trace.py <..._kernel_trace.csv> [launches of the timed region, default 3000]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tower" in r["Kernel_Name"]]
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 3000):]
ev = []
for r in rows:
    ev += [(int(r["Start_Timestamp"]), 1), (int(r["End_Timestamp"]), -1)]
ev.sort()
cur, last, acc = 0, ev[0][0], {}
for t, dv in ev:
    acc[cur] = acc.get(cur, 0) + t - last
    last, cur = t, cur + dv
total = float(sum(acc.values()))
print("k_tower launches in flight over the last %d launches (%.1f ms, queues %s): %s" % (
    len(rows), total / 1e6, sorted(set(r["Queue_Id"] for r in rows)),
    "  ".join("%d: %.1f %%" % (k, 100 * v / total) for k, v in sorted(acc.items()))))
