#!/usr/bin/env python3
"""Headline benchmark: batched MCTS self-play on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one search iteration of the hot path over one batch: every one of the
`games` concurrent trees descends by PUCT, expands one leaf, the fused conv tower
evaluates the leaf batch, and the results are backed up (plus move sampling / re-rooting /
game turnover for the trees whose root reached `visits`).  Metric (BASELINE.json):
MCTS node-evals/s at 400 sims/move; games/s reported beside it.  Workload at N=1: 4096
concurrent games, 400 sims/move, 12x128 net (random-init seed 1, .npy layout), bf16 — the
per-GPU shard of BASELINE.json configs[2] (32768 games over 8 GPUs), i.e. configs[1]'s game
count at the sims/move the metric is quoted on.  N ranks = N x 4096 games, weak scaling, no
data-path collective.

Prints ONE JSON line on rank 0, with the dominant kernel's roofline (MFMA, measured with
HIP events on the engine's stream) and a CPU baseline (the oracle port on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0


def tree_bytes(d, logit_bytes=4):
    """Algorithmic bytes of the tree kernels: per step 16L + 12C (select) +
    24L (backup) + 32 + 18M (expand) + 834e (policy row + value)."""
    return (40 * d["levels"] + 12 * d["children"] + 32 * d["steps"] + 18 * d["new_moves"] +
            834 * logit_bytes * d["nn_evals"])


def measured_traffic():
    """HBM bytes per tower launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE, doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950, + WRITE_SIZE), as summarised by tools/prof_bench.sh into
    profiles/.  PMC counters cannot be read from inside the process, so this is the committed measurement, or null."""
    import re
    path = os.path.join(ROOT, "profiles", "round1_bench_pmc_k_tower.txt")
    try:
        m = re.search(r"x2 corrected: ([0-9.e+]+) MB\), WRITE_SIZE [0-9.e+]+ KiB \(([0-9.e+]+) MB\)", open(path).read())
        return (float(m.group(1)) + float(m.group(2))) * 1e6, os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def cpu_baseline(conv, bn, visits, seconds=15.0, games=128):
    """The oracle port (sequential PUCT per game + numpy f32 net) on the host cores, on a
    bounded sample of the same workload: `games` concurrent games at the same sims/move."""
    import numpy as np
    from oracle import net_oracle
    from oracle import oracle_lib as orc
    cfg = orc.make_config(games=games, visits=visits, seed=20260101)
    eng = orc.Engine(cfg)
    logits = np.zeros((games, 833), np.float32)
    values = np.zeros(games, np.float32)
    t0 = time.time()
    iters = 0
    while time.time() - t0 < seconds:
        n, need = eng.select()
        lb = eng.leaf_boards()
        idx = np.nonzero(need)[0]
        if len(idx):
            p, v = net_oracle.forward(conv, bn, net_oracle.features_from_leaf_boards(lb[idx], cfg.blockers, np.float32),
                                      dtype=np.float32)
            logits[idx] = p.reshape(len(idx), 833)
            values[idx] = v.reshape(len(idx))
        eng.backup(logits, values)
        iters += 1
    dt = time.time() - t0
    st = eng.stats()
    # the tree side alone (null evaluator: zero logits, zero value), as SURVEY 8(d) asks beside the full figure
    eng2 = orc.Engine(orc.make_config(games=games, visits=visits, seed=20260101))
    zl, zv = np.zeros((games, 833), np.float32), np.zeros(games, np.float32)
    t2 = time.time()
    while time.time() - t2 < min(3.0, seconds):
        eng2.select()
        eng2.backup(zl, zv)
    tree_only = eng2.stats()["steps"] / (time.time() - t2)
    return {"value": st["steps"] / dt, "unit": "node-evals/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
            "tree_only_value": tree_only,
            "sample": "%d concurrent games x %d iterations (%.1f s): oracle tree search (1 thread) + numpy f32 "
                      "conv tower (BLAS threads = host cores), same net / sims-per-move; tree_only_value = the same "
                      "search with a null evaluator, 1 thread" % (games, iters, dt)}


def target_leg(conv, bn, args, games=16384, steps=300, warmup=100):
    """BASELINE.json's north-star operating point (>= 10k concurrent games on one GPU, 400 sims/move) measured
    the same way as the headline, reported beside it (never as `value`)."""
    from ataxxzero_amd import model, selfplay
    sp = selfplay.SelfPlay(conv, bn, games=games, visits=args.visits, dtype=args.dtype, seed=selfplay.DEFAULT_SEED + 77,
                           select_budget=args.select_budget)
    def run(iters):
        done = 0
        while done < iters:
            n = min(args.chunk, iters - done)
            sp.run(n)
            sp.drain()
            done += n

    try:
        sp.set_visits(min(16, args.visits))
        run(args.phase_mix)
        sp.set_visits(args.visits)
        run(300)
        run(warmup)
        sp.sync()
        st0 = sp.stats()
        sp.timing_reset(True)
        t0 = time.perf_counter()
        run(steps)
        sp.sync()
        dt = time.perf_counter() - t0
        st1 = sp.stats()
        tm = sp.timing()
        it = max(tm["iterations"], 1)
        evals = (st1["nn_evals"] - st0["nn_evals"]) * it / float(steps)
        tf = evals * model.flops_per_eval(args.blocks, 128) / (tm["net_ms"] * 1e-3) / 1e12
        return {"games": games, "node_evals_per_s": (st1["steps"] - st0["steps"]) / dt, "ms_per_step": 1e3 * dt / steps,
                "steps": steps,
                "tower_ms_per_launch": tm["net_ms"] / it, "tower_tflops": tf, "tower_frac_of_peak": tf / MFMA_PEAK_TFLOPS[args.dtype],
                "tree_ms_per_step": (tm["select_ms"] + tm["backup_ms"]) / it}
    finally:
        sp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--chunk", type=int, default=250, help="iterations between finished-game drains")
    ap.add_argument("--streams", type=int, default=1, help="half-batches in flight per GPU (the reference's double buffer)")
    ap.add_argument("--phase-mix", type=int, default=2500,
                    help="untimed setup iterations at 16 sims/move that spread the games over all game phases "
                         "(a fresh start has every game at ply 0 with an empty tree), followed by --phase-fill "
                         "iterations at full sims/move that regrow the trees; then the --warmup steps")
    ap.add_argument("--phase-fill", type=int, default=500)
    ap.add_argument("--select-budget", type=int, default=48,
                    help="tree levels per select launch and game (azh_config.select_budget; 0 = unlimited): deeper "
                         "descents park and resume next iteration, so a launch does not last as long as the deepest "
                         "line of the batch; every game still plays exactly the same search")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-leg", action="store_true", help="skip the 16384-game leg reported beside the headline")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    from ataxxzero_amd import distrib, link, model, selfplay
    group = distrib.Group()
    if group.world != args.gpus and group.world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, group.world))
    ndev = link.require_gpu()
    link.check(link.load().azh_set_device(distrib.device_for(group.local_rank, ndev)))

    conv, bn = model.random_init(args.blocks, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=args.games, visits=args.visits, dtype=args.dtype,
                           seed=distrib.shard_seed(selfplay.DEFAULT_SEED, group.rank), streams=args.streams,
                           select_budget=args.select_budget)

    def run(iters):
        done = 0
        finished = 0
        while done < iters:
            n = min(args.chunk, iters - done)
            sp.run(n)
            finished += len(sp.drain())   # sync + hand finished games to the host, as the CLI does
            done += n
        return finished

    if args.phase_mix > 0:
        sp.set_visits(min(16, args.visits))
        run(args.phase_mix)
        sp.set_visits(args.visits)
        run(args.phase_fill)
    run(args.warmup)
    sp.sync()
    group.barrier()
    st0 = sp.stats()
    sp.timing_reset(True)
    t0 = time.perf_counter()
    finished = run(args.steps)
    sp.sync()
    t1 = time.perf_counter()
    group.barrier()
    st1 = sp.stats()
    tm = sp.timing()
    d = {k: st1[k] - st0[k] for k in st1}
    steps_total, t_max, rate = distrib.aggregate(group, d["steps"], t1 - t0)
    evals_total = group.reduce(d["nn_evals"], "sum")
    plies_total = group.reduce(d["plies"], "sum")
    games_total = group.reduce(finished, "sum")

    if group.rank == 0:
        flops = model.flops_per_eval(args.blocks, 128)
        it = max(tm["iterations"], 1)   # launches recorded (per half-batch when streams > 1)
        frac_timed = it / float(args.steps * args.streams)
        net_s = tm["net_ms"] * 1e-3
        achieved_tf = d["nn_evals"] * frac_timed * flops / net_s / 1e12 if net_s > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        traffic, traffic_src = measured_traffic()
        tree_s = (tm["select_ms"] + tm["backup_ms"]) * 1e-3
        tree_gbs = tree_bytes(d) * frac_timed / tree_s / 1e9 if tree_s > 0 else 0.0
        out = {
            "metric": "MCTS node-evals/s at %d sims/move (self-play; games/s beside it)" % args.visits,
            "value": rate, "unit": "node-evals/s", "n_gpus": group.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init .npy-layout weights, seed 1; "
                                                              "self-play from the reference start position)",
            "config": {"workload": "%d concurrent self-play games per GPU, %d sims/move, %dx128 conv net, %s "
                                   "(per-GPU shard of BASELINE configs[2]; configs[1] at the metric's 400 sims)"
                                   % (args.games, args.visits, args.blocks, args.dtype),
                       "games_per_gpu": args.games, "half_batches_in_flight": args.streams, "visits": args.visits,
                       "select_budget": args.select_budget,
                       "setup": "games spread over all phases by %d untimed 16-sim iterations + %d at full sims, then "
                                "the warm-up" % (args.phase_mix, args.phase_fill if args.phase_mix > 0 else 0),
                       "net": "%dx128" % args.blocks,
                       "parallelism": "%d independent game shards, no collective" % group.world},
            "nn_evals_per_s": evals_total / t_max, "plies_per_s": plies_total / t_max,
            # finished games per second in steady state = plies/s / mean game length; 147.75 plies is the mean
            # over 29,296 games written by the real CLI in a 420 s soak of this workload
            "games_per_s": (plies_total / t_max) / 147.75,
            "games_finished_in_timed_region": games_total,
            "roofline": {"bound": "mfma", "kernel": "%s<%s>" % ("k_tower" if args.dtype == "f32" or os.environ.get("AZH_TOWER") == "1" else "k_tower2", args.dtype), "achieved": achieved_tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved_tf / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": tm["net_ms"] / it, "evals_per_launch": d["nn_evals"] / float(args.steps * args.streams),
                         "flops_per_eval": flops},
            "tree_roofline": {"bound": "hbm", "kernels": "k_tree (backup + move-due mark + select, fused) + k_compact on the engine's stream; "
                                         "k_advance_list (re-roots) runs on a side stream under the tower",
                              "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tree_gbs / HBM_PEAK_GBS,
                              "select_ms_per_step": tm["select_ms"] / it, "backup_ms_per_step": tm["backup_ms"] / it,
                              "bytes_per_step": tree_bytes(d) / float(max(d["steps"], 1)),
                              "levels_per_step": d["levels"] / float(max(d["steps"], 1)),
                              "children_per_step": d["children"] / float(max(d["steps"], 1))},
            "counters": d,
        }
        sp.close()
        if group.world == 1 and not args.no_target_leg and args.streams == 1:
            out["target_10k_games"] = target_leg(conv, bn, args)
        if group.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(conv, bn, args.visits, seconds=args.cpu_seconds)
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        sp.close()
    group.close()


if __name__ == "__main__":
    main()
