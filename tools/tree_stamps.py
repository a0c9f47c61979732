#!/usr/bin/env python3
"""Where the time of the fused tree launch (k_tree) goes, from in-kernel s_memrealtime stamps (diagnostic instantiation).

The engine is put in bench.py's steady state (positions of profiles/round2_steady_state_positions.npz, trees grown),
then `--samples` stamped launches are taken, a few ordinary iterations apart.  Per game wave the stamps are: wave start,
state loaded, backup done, move-due mark done, descent done, expansion done, state stored, workgroup (its four games; one
beyond 8192 games) done.
Printed: per phase the mean / median / p90 / max over all waves in microseconds, the start skew of the waves (dispatch),
and the span of the whole launch (first wave start to last workgroup done), i.e. what the iteration waits for.

    python tools/tree_stamps.py [--games 4096] [--visits 400] [--select-budget 48] [--eval-cache] [--samples 8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ataxxzero_amd import link, model, selfplay  # noqa: E402

PHASES = ["load state + force flag", "backup (priors, noise, path update)", "move-due mark", "descent (PUCT levels)",
          "expansion (makemove, movegen, edges)", "store state / flag / leaf board"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--select-budget", type=int, default=48)
    ap.add_argument("--eval-cache", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--fill", type=int, default=1000)
    args = ap.parse_args()
    link.require_gpu()
    conv, bn = model.random_init(args.blocks, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=args.games, visits=args.visits, dtype=args.dtype, select_budget=args.select_budget,
                           flags=link.FLAG_EVAL_CACHE if args.eval_cache else 0)
    snap = np.load(os.path.join(ROOT, "profiles", "round2_steady_state_positions.npz"))
    rng = np.random.default_rng(1)
    pick = rng.permutation(len(snap["plies"])) if args.games == len(snap["plies"]) else rng.integers(0, len(snap["plies"]), args.games)
    sp.set_positions(snap["boards"][pick], snap["plies"][pick])
    done = 0
    while done < args.fill:
        sp.run(250)
        sp.drain()
        done += 250
    e = sp.engine
    rows, spans, skews, wg_waits, levels, per_level = [], [], [], [], [], []
    for _ in range(args.samples):
        st0 = e.stats()
        t = e.tree_stamps(sp.net, link.DTYPES[args.dtype]).astype(np.int64)
        st1 = e.stats()
        sp.run(37)
        t = t[t[:, 0] > 0]
        rows.append(np.diff(t[:, :7], axis=1) / 100.0)          # 100 MHz ticks -> microseconds
        spans.append((t[:, 7].max() - t[:, 0].min()) / 100.0)
        skews.append((t[:, 0] - t[:, 0].min()) / 100.0)
        wg_waits.append((t[:, 7] - t[:, 6]) / 100.0)
        deep = t[:, 8] >= 8
        per_level.append(np.stack([(t[deep, 4] - t[deep, 3]) / 100.0 / t[deep, 8], t[deep, 8], t[deep, 9] / t[deep, 8],
                                   (t[deep, 3] - t[:, 0].min()) / 100.0], axis=1))
        # two iterations ran; the stamped one is the first's tree phase: levels per step over both, as context
        levels.append((st1["levels"] - st0["levels"]) / max(1, st1["steps"] - st0["steps"]))
    d = np.concatenate(rows)
    print("k_tree stamps: %d games, %d sims/move, select budget %d%s, %d stamped launches in steady state" % (
        args.games, args.visits, args.select_budget, ", eval cache" if args.eval_cache else "", args.samples))
    print("levels per step around the samples: %.1f" % np.mean(levels))
    print("%-42s %8s %8s %8s %8s   (us per game wave)" % ("phase", "mean", "p50", "p90", "max"))
    for k, name in enumerate(PHASES):
        c = d[:, k]
        print("%-42s %8.2f %8.2f %8.2f %8.2f" % (name, c.mean(), np.median(c), np.percentile(c, 90), c.max()))
    total = d.sum(axis=1)
    print("%-42s %8.2f %8.2f %8.2f %8.2f" % ("one game, start to state stored", total.mean(), np.median(total),
                                              np.percentile(total, 90), total.max()))
    pl = np.concatenate(per_level)
    print("descents of >= 8 levels: %.3f us per level on average (p50 %.3f, p90 %.3f); %.1f levels, %.1f children per level" % (
        pl[:, 0].mean(), np.median(pl[:, 0]), np.percentile(pl[:, 0], 90), pl[:, 1].mean(), pl[:, 2].mean()))
    for lo, hi in ((8, 16), (16, 32), (32, 47), (47, 1000)):
        m = (pl[:, 1] >= lo) & (pl[:, 1] < hi)
        if m.any():
            print("   %3d-%-4d levels: %6d descents, %.3f us per level" % (lo, hi, m.sum(), pl[m, 0].mean()))
    sk = np.concatenate(skews)
    print("%-42s %8.2f %8.2f %8.2f %8.2f" % ("wave start after the launch's first wave", sk.mean(), np.median(sk),
                                              np.percentile(sk, 90), sk.max()))
    ww = np.concatenate(wg_waits)
    print("%-42s %8.2f %8.2f %8.2f %8.2f" % ("waiting for the workgroup's other games", ww.mean(), np.median(ww),
                                              np.percentile(ww, 90), ww.max()))
    print("launch span, first wave start -> last workgroup done: mean %.1f us  min %.1f  max %.1f  (the iteration waits for "
          "this, plus the compaction by the last workgroup and the kernel boundary)" % (np.mean(spans), np.min(spans), np.max(spans)))
    sp.close()


if __name__ == "__main__":
    main()
