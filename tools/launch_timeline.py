#!/usr/bin/env python3
"""Where a tower launch spends its time: start / end stamps of every workgroup (diagnostic s_memtime build) —
the dispatch ramp, the duration of first-round workgroups (all in lock-step) against later ones, the drain at the end.
usage (GPU box): python tools/launch_timeline.py 1536 3072 3600 4608"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ataxxzero_amd import link, model

BLOCKS = int(os.environ.get("BLOCKS", "12"))
conv, bn = model.random_init(BLOCKS, 128, seed=1)
net = link.Net(conv, bn)
for n in [int(a) for a in sys.argv[1:]] or [1536, 3600]:
    wgs = (n + 2) // 3
    st = net.stamps(n, wgs=wgs).astype(np.int64)            # [wg][wave][128]
    t0 = st[:, :, 0].min(axis=1)
    te = st[:, :, 2].max(axis=1)
    # s_memtime has its own base on every XCD; workgroup i runs on XCD i % 8, and the first workgroup of every XCD
    # starts within a microsecond of the others
    for x in range(8):
        base = t0[x::8].min()
        t0[x::8] -= base
        te[x::8] -= base
    dur = te - t0
    order = np.argsort(t0)
    first = order[:512]
    later = order[512:]
    print("n=%d boards, %d workgroups: span %.0f kcycles (first start -> last end)" % (n, wgs, te.max() / 1e3))
    print("  starts of the first 512: p50 %.1f  p90 %.1f  max %.1f kcycles" % tuple(np.percentile(t0[first], [50, 90, 100]) / 1e3))
    print("  durations, first 512: p10 %.0f p50 %.0f p90 %.0f kcycles" % tuple(np.percentile(dur[first], [10, 50, 90]) / 1e3))
    if len(later):
        print("  durations, later %d: p10 %.0f p50 %.0f p90 %.0f kcycles" % ((len(later),) + tuple(np.percentile(dur[later], [10, 50, 90]) / 1e3)))
        print("  starts of the later ones: p10 %.0f p50 %.0f p90 %.0f; ends of all: p50 %.0f p90 %.0f p99 %.0f max %.0f kcycles" % (
            tuple(np.percentile(t0[later], [10, 50, 90]) / 1e3) + tuple(np.percentile(te, [50, 90, 99, 100]) / 1e3)))
    # per-layer loop time by start cohort (wave 0, residual layers 1..24)
    L = np.stack([st[:, 0, 8 + 4 * i: 12 + 4 * i] for i in range(2 * BLOCKS)], axis=1)   # [wg][layer][4]
    per_layer = L[:, :, 3] - L[:, :, 0]
    print("  layer time (loop start -> barrier passed), first 512 by layer thirds: %s kcycles" % np.round(
        [per_layer[first][:, a:a + 8].mean() / 1e3 for a in range(0, 2 * BLOCKS, 8)], 2))
    if len(later):
        print("  same, later workgroups: %s kcycles" % np.round([per_layer[later][:, a:a + 8].mean() / 1e3 for a in range(0, 2 * BLOCKS, 8)], 2))
    for name, grp in (("first 512", first), ("started after 1/3 of the span", order[t0[order] > te.max() / 3]), ("started after 2/3", order[t0[order] > 2 * te.max() / 3])):
        if len(grp):
            print("  per layer 1..24, %s (%d): %s" % (name, len(grp), " ".join("%.0f" % (v / 1e3) for v in per_layer[grp].mean(axis=0))))
    # busy workgroups over time (how long the machine runs below 512 resident workgroups)
    ts = np.linspace(0, te.max(), 41)
    occ = [(int(((t0 <= t) & (te > t)).sum())) for t in ts]
    print("  resident workgroups at 40 equal time steps: %s" % occ)

# ---- second pass with the constant-rate clock (s_memrealtime, 100 MHz, one base for the whole chip)
print()
print("real-time view (s_memrealtime, 10 ns units): when workgroups run and what the shader clock is while they do")
for n in [int(a) for a in sys.argv[1:]] or [1536, 3600]:
    wgs = (n + 2) // 3
    st = net.stamps(n, wgs=wgs).astype(np.int64)
    r0 = st[:, 0, 3]
    r1 = st[:, 0, 104]
    base = r0.min()
    span_us = (r1.max() - base) / 100.0
    starts = (r0 - base) / 100.0
    ends = (r1 - base) / 100.0
    print("n=%d: span %.0f us; starts p50 %.0f p90 %.0f max %.0f us; workgroup life p10 %.0f p50 %.0f p90 %.0f us" % (
        (n, span_us) + tuple(np.percentile(starts, [50, 90, 100])) + tuple(np.percentile(ends - starts, [10, 50, 90]))))
    ts = np.linspace(0, span_us, 21)
    print("  resident workgroups every %.0f us: %s" % (ts[1], [int(((starts <= t) & (ends > t)).sum()) for t in ts]))
    # clock: shader cycles between the ends of consecutive residual blocks over the real time between them
    cyc = np.stack([st[:, 0, 12 + 8 * b + 3] for b in range(BLOCKS)], axis=1)
    rt = np.stack([st[:, 0, 105 + b] for b in range(BLOCKS)], axis=1)
    ghz = np.diff(cyc, axis=1) / (np.diff(rt, axis=1) * 10.0)       # cycles per ns
    mid = ((rt[:, 1:] + rt[:, :-1]) / 2 - base) / 100.0
    bins = np.linspace(0, span_us, 13)
    out = []
    for a, b in zip(bins[:-1], bins[1:]):
        m = (mid >= a) & (mid < b)
        out.append("%.2f" % ghz[m].mean() if m.any() else "-")
    print("  shader clock (GHz) in 12 equal slices of the span: %s" % " ".join(out))
