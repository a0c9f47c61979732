#!/usr/bin/env python3
"""What the host round trip between two runs of the device loop costs (bench.py's step = 250 iterations + one drain;
the generator CLI does the same): per step, the time to enqueue the run, the wait for the GPU, and — after that wait, with
the GPU idle — the drain (ring copy, JSON formatting of the finished games) and the counter read.

    python tools/drain_cost.py [--games 4096] [--steps 12]     # on the GPU box, from the repo root
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from ataxxzero_amd import link, model, selfplay  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--iters", type=int, default=250)
    ap.add_argument("--eval-cache", action="store_true")
    a = ap.parse_args()
    link.require_gpu()
    conv, bn = model.random_init(12, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=a.games, visits=a.visits, dtype="bf16", seed=selfplay.DEFAULT_SEED,
                           select_budget=64 if a.eval_cache else 48, flags=link.FLAG_EVAL_CACHE if a.eval_cache else 0)
    bench.spread(sp, types.SimpleNamespace(phase_fill=1000), selfplay.DEFAULT_SEED)
    for _ in range(2):
        sp.run(a.iters)
        sp.drain()
    sp.sync()
    rows = []
    games0 = sp.stats()["games"]
    t_all = time.perf_counter()
    for _ in range(a.steps):
        t0 = time.perf_counter()
        sp.run(a.iters)
        t1 = time.perf_counter()
        sp.sync()
        t2 = time.perf_counter()
        lines = sp.drain()
        t3 = time.perf_counter()
        st = sp.stats()
        t4 = time.perf_counter()
        rows.append({"enqueue_ms": 1e3 * (t1 - t0), "gpu_wait_ms": 1e3 * (t2 - t1), "drain_ms": 1e3 * (t3 - t2),
                     "stats_ms": 1e3 * (t4 - t3), "lines": len(lines), "bytes": sum(len(x) for x in lines),
                     "games_finished": st["games"] - games0})
        games0 = st["games"]
    total = time.perf_counter() - t_all
    mean = lambda k: sum(r[k] for r in rows) / len(rows)
    out = {"games": a.games, "visits": a.visits, "iterations_per_step": a.iters, "steps": a.steps, "eval_cache": a.eval_cache,
           "ms_per_step": 1e3 * total / a.steps,
           "mean": {k: round(mean(k), 3) for k in ("enqueue_ms", "gpu_wait_ms", "drain_ms", "stats_ms", "lines", "bytes",
                                                   "games_finished")},
           "host_gap_share": (mean("drain_ms") + mean("stats_ms")) / (1e3 * total / a.steps),
           "rows": [{k: round(v, 3) for k, v in r.items()} for r in rows]}
    print(json.dumps(out))
    sp.close()


if __name__ == "__main__":
    main()
