#!/usr/bin/env python3
"""Headline benchmark: batched MCTS self-play on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload at N=1: 4096 concurrent games, 400 sims/move, 12x128 net (random-init seed 1, .npy layout), bf16 — the
per-GPU shard of BASELINE.json configs[2] (32768 games over 8 GPUs), i.e. configs[1]'s game count at the sims/move
the metric is quoted on.  N ranks = N x 4096 games, weak scaling, no data-path collective.

A STEP is one round trip of the generator's host loop (accelerated_generate_games.py:54-83 drains finished games
once per round trip): ITERS_PER_STEP = 250 search iterations of the hot path over the whole batch — in every
iteration each of the `games` trees descends by PUCT and expands one leaf, the fused conv tower evaluates the leaf
batch, the results are backed up, due moves are sampled / re-rooted / recorded — followed by one drain of the
finished games' records to the host.  Metric (BASELINE.json): MCTS node-evals/s at 400 sims/move; games/s =
finished games drained in the timed region / its wall time.  The games are first spread over all game phases
(a fresh start has every game at ply 0), so both figures are steady-state figures.

Prints ONE JSON line on rank 0, with the dominant kernel's roofline (MFMA; HIP events on the engine's stream,
sampled over the timed region) and a CPU baseline: the reference's architecture (one host thread per game,
double-buffered batches) on this box's cores with the GPU tower as its evaluator.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, one child process per
GPU, before anything touches the GPU in this process; under torchrun it is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0
SCATTERED_PEAK_GBS = 5300.0   # dependent scattered 672-B reads, >= 8192 chains in flight, measured (profiles/round3_random_chase.txt)
ITERS_PER_STEP = 250
TIMING_STRIDE = 8        # every 8th iteration is event-timed (an event is a barrier packet in the queue)
# rocprofv3 PMC summaries of this very command (tools/prof_bench.sh), per launch of the tower: by half-batches in flight
PMC_SUMMARY = {2: os.path.join("profiles", "round6_prof_default_two_halves.txt"),
               1: os.path.join("profiles", "round6_prof_default_one_batch.txt")}
KERNEL_SOURCE = os.path.join("ataxxzero_amd", "csrc", "net_kernels.hip")
SNAPSHOT_MEAN_GAME_PLIES = 152.42   # mean length of the 4096 games of the generation the steady-state snapshot was drawn from
                                    # (profiles/round2_steady_state_positions.npz, meta.generation0_mean_plies; 400 sims/move)


def tree_bytes(d, logit_bytes=4):
    """Algorithmic bytes of the tree kernels: per step 16L + 12C (select) +
    24L (backup) + 32 + 18M (expand) + 834e (policy row + value)."""
    return (40 * d["levels"] + 12 * d["children"] + 32 * d["steps"] + 18 * d["new_moves"] +
            834 * logit_bytes * d["nn_evals"])


def source_sha256(rel=KERNEL_SOURCE):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
    except OSError:
        return None


def measured_traffic(streams):
    """HBM bytes per tower launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE, doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950, + WRITE_SIZE), as summarised by tools/prof_bench.sh into
    profiles/.  PMC counters cannot be read from inside the process, so this is the committed measurement — and only while
    it is a measurement of THIS kernel: the summary records the sha256 of the kernel source it was taken with, and a
    summary of another source (or of none) gives null."""
    import re
    rel = PMC_SUMMARY.get(streams)
    if rel is None:
        return None, None
    try:
        text = open(os.path.join(ROOT, rel)).read()
        h = re.search(r"kernel source sha256: ([0-9a-f]{64})", text)
        if not h or h.group(1) != source_sha256():
            return None, "%s is not a measurement of the current %s (sha256 differs): stale, not reported" % (rel, KERNEL_SOURCE)
        m = re.search(r"x2 corrected: ([0-9.e+]+) MB\), WRITE_SIZE [0-9.e+]+ KiB \(([0-9.e+]+) MB\)", text)
        return (float(m.group(1)) + float(m.group(2))) * 1e6, rel
    except Exception:
        return None, None


def cpu_baseline(net, visits, dtype, seconds):
    """The reference's architecture on this box (tools/cpu_baseline): 2B host game threads doing sequential PUCT,
    B-row double buffer, the host loop handing each full buffer to an evaluator.  Three legs on a bounded sample:
    (ii) the GPU tower as evaluator on all host cores = `value`; (i) a null evaluator on all cores and pinned to 8
    cores — the figure BASELINE.md §2 measured from the real cpp/self_play_client.so (38-49 k steps/s)."""
    import numpy as np
    from ataxxzero_amd import build, link, selfplay
    from tools.cpu_baseline import driver
    build.build_cpu_baseline()
    blockers = selfplay.parse_fen(selfplay.START_FEN_SELFPLAY)[2]
    dt = link.DTYPES[dtype]
    cores = sorted(os.sched_getaffinity(0))
    B = 256

    def tower(boards):
        p, v = net.forward(boards, blockers, dt)
        return p.reshape(len(boards), 833), v.reshape(len(boards))

    # the evaluator leg gets the buffer size that suits it best: a larger batch amortises the per-call cost of the
    # host-buffer evaluator (the reference's default is 128 rows, accelerated_generate_games.py:20)
    # (at most 896 game threads: the GPU boxes allow about a thousand tasks per job)
    legs = [driver.run(tower, visits, b, seconds / 2.0) for b in (256, 448)]
    full = max(legs, key=lambda r: r["steps_per_s"])
    B = full["buffer_entries"]
    null_all = driver.run(None, visits, 256, min(seconds, 5.0))
    null_8 = None
    if len(cores) > 8:
        os.sched_setaffinity(0, set(cores[:8]))   # the game threads inherit it
        try:
            null_8 = driver.run(None, visits, 64, min(seconds, 5.0))
        finally:
            os.sched_setaffinity(0, set(cores))
    return {"value": full["steps_per_s"], "unit": "node-evals/s", "cores": len(cores), "kind": "port",
            "nn_evals_per_s": full["evals_per_s"], "plies_per_s": full["plies_per_s"],
            "evaluator_legs": {str(r["buffer_entries"]): r["steps_per_s"] for r in legs},
            "null_evaluator_value": null_all["steps_per_s"],
            "null_evaluator_8_cores_value": (null_8 or null_all)["steps_per_s"],
            "reference_so_null_evaluator_8_cores": "38-49 k steps/s (BASELINE.md §2, the real cpp/self_play_client.so)",
            "sample": "%d host game threads (buffer %d, the reference's 2B threads / B-row double buffer), %d sims/move, "
                      "%.1f s timed after a 1 s warm-up; evaluator = this GPU's %s tower through azh_net_forward "
                      "(host buffers: PCIe both ways, as the reference's sess.run); null legs: all-zero evaluations"
                      % (full["threads"], B, visits, full["seconds"], dtype)}


def vendor_gemm_ceiling():
    """Dense bf16 GEMM through the vendor library (torch.matmul -> hipBLASLt) on random data, on this box, right after the
    timed region: what an MFMA-dense kernel reaches here under the chip's power limit (the 2.5 PFLOP/s peak assumes
    2.4 GHz).  Context for roofline.frac, never a replacement for `peak`.  Runs in a child process: torch brings its own
    HIP runtime, which must not be initialised after this process's."""
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_ceiling.py"), "--json"],
                             capture_output=True, timeout=300)   # (a first `import torch` on a fresh box can take a minute)
        return json.loads(res.stdout.decode().strip().splitlines()[-1])
    except Exception as e:   # context only: never fail the bench for it
        return {"error": repr(e)}


def measure(sp, args, steps, warmup, group=None):
    """`warmup` untimed steps, then EXACTLY `steps` timed steps between barriers + device syncs."""
    per_step = []
    seen = [sp.stats()["games"]]

    def run_steps(n):
        written = 0
        for _ in range(n):
            sp.run(args.iters_per_step)
            written += len(sp.drain())    # sync + hand the finished games to the host, as the CLI does
            now = sp.stats()["games"]     # games finished so far (device counter)
            per_step.append(now - seen[0])
            seen[0] = now
        return written

    run_steps(warmup)
    sp.sync()
    if group is not None:
        group.barrier()
    st0 = sp.stats()
    sp.timing_reset(TIMING_STRIDE)
    t0 = time.perf_counter()
    finished = run_steps(steps)
    sp.sync()
    t1 = time.perf_counter()
    if group is not None:
        group.barrier()
    st1 = sp.stats()
    return {k: st1[k] - st0[k] for k in st1}, finished, t1 - t0, sp.timing(), per_step[warmup:]


SNAPSHOT = os.path.join("profiles", "round2_steady_state_positions.npz")


def spread(sp, args, seed):
    """Untimed set-up: a fresh start has every game at ply 0 with an empty tree — nothing like the state the generator
    works in, and the steady state takes several game generations (one is about 70 s) to form.  The slots are therefore
    LOADED with steady-state positions built from one complete generation of real games of this very workload
    (profiles/round2_steady_state_positions.npz, tools/steady_state_positions.py; sampled with replacement when the
    batch is not 4096 games), azh_engine_set_positions; --phase-fill iterations at full sims/move then grow the trees.
    Games that start from a loaded position are played and counted, not written.  `--phase-fill 0` starts cold."""
    if args.phase_fill <= 0:
        return None
    import numpy as np
    snap = np.load(os.path.join(ROOT, SNAPSHOT))
    boards, plies = snap["boards"], snap["plies"]
    rng = np.random.default_rng(seed)   # per-rank seed: every shard gets its own assignment of positions to slots
    pick = rng.permutation(len(plies)) if sp.games == len(plies) else rng.integers(0, len(plies), size=sp.games)
    sp.set_positions(boards[pick], plies[pick])
    done = 0
    while done < args.phase_fill:
        sp.run(min(250, args.phase_fill - done))
        sp.drain()
        done += 250
    return {"mean_ply": float(plies[pick].mean()), "source": SNAPSHOT}


def target_leg(conv, bn, args, games=16384, steps=8, warmup=1, flags=0, dtype=None, select_budget=None, streams=1,
               visits=None, blocks=None):
    """Another operating point measured the same way as the headline and reported beside it (never as `value`):
    BASELINE.json's north-star point (>= 10k concurrent games on one GPU, 400 sims/move), the headline workload with the
    evaluation cache on (the generator CLI's default), with the f16 tower, or another BASELINE config.
    In EVERY leg `tower_frac_of_peak` is the tower's rate over the chip — all its FLOPs in the leg's timed region / the
    region's wall time, the definition of the headline's roofline.frac — and `per_launch_frac` is one launch's own rate
    (FLOPs per launch / its average duration by HIP events: with two half-batches in flight it prices about half a chip)."""
    from ataxxzero_amd import model, selfplay
    dtype = dtype or args.dtype
    visits = visits or args.visits
    blocks = blocks or args.blocks
    budget = args.select_budget if select_budget is None else select_budget
    sp = selfplay.SelfPlay(conv, bn, games=games, visits=visits, dtype=dtype, seed=selfplay.DEFAULT_SEED + 77,
                           select_budget=budget, flags=flags, streams=streams)
    try:
        spread(sp, args, selfplay.DEFAULT_SEED + 77)
        d, finished, dt, tm, _ = measure(sp, args, steps, warmup)
        it = max(tm["iterations"], 1)
        iters = steps * args.iters_per_step
        flops = model.flops_per_eval(blocks, 128)
        launch_tf = d["nn_evals"] / float(iters * streams) * flops / (tm["net_ms"] / it * 1e-3) / 1e12
        region_tf = d["nn_evals"] * flops / dt / 1e12
        tree_ms = (tm["select_ms"] + tm["backup_ms"]) / it
        tree_gbs = tree_bytes(d) / float(iters * streams) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
        return {"games": games, "visits": visits, "net": "%dx128" % blocks, "dtype": dtype, "eval_cache": bool(flags),
                "select_budget": budget, "half_batches_in_flight": streams,
                "node_evals_per_s": d["steps"] / dt, "nn_evals_per_s": d["nn_evals"] / dt,
                "cache_hits_per_s": d.get("cache_hits", 0) / dt, "plies_per_s": d["plies"] / dt, "games_per_s": d["games"] / dt,
                "ms_per_iteration": 1e3 * dt / iters, "steps": steps,
                "tower_tflops": region_tf, "tower_frac_of_peak": region_tf / MFMA_PEAK_TFLOPS[dtype],
                "per_launch_tflops": launch_tf, "per_launch_frac": launch_tf / MFMA_PEAK_TFLOPS[dtype],
                "evals_per_launch": d["nn_evals"] / float(iters * streams), "tower_ms_per_launch": tm["net_ms"] / it,
                "tree_ms_per_iteration": tree_ms, "tree_roofline_frac": tree_gbs / HBM_PEAK_GBS,
                "parked_share": d.get("parked", 0) / float(max(1, d.get("parked", 0) + d["steps"]))}
    finally:
        sp.close()


def config5_leg(games=1000, visits=100, dtype="f16", seed=None, net_seeds=(1, 2)):
    """BASELINE configs[4] (uai_ringmaster.py with two uai_interface.py engines, :75-160,221-265) through this repo's batched
    arena: a FIXED cohort of `games` games — every pairing both ways — between two random-init 12x128 nets (`net_seeds`),
    `visits` MCTS steps per move from a fresh tree, played to completion as uai_ringmaster.py's drop-in plays it (50 search
    iterations per round trip, finished games drained and scored; thin batches — one board per workgroup — once at most 512
    games of the match are left).  What the figures mean: `wall_s` is the longest game of the cohort at the latency of a
    thinning batch — with one game cut at 400 plies most of it prices a near-empty batch — so the leg also says when 99 %
    of the games were over (`wall_s_until_99pct_games`, `iterations_until_99pct`), the rates while the batch was still a batch
    (`..._while_ge512_live`: up to the fetch after which at most 512 games were left = the switch to thin batches), and one
    tower launch's own rate from HIP events (`per_launch_frac`, every 8th launch; over the whole match and over that
    first phase)."""
    from ataxxzero_amd import arena, model, selfplay
    ROUND = 50   # search iterations per host round trip (uai_ringmaster.py's drop-in uses the same)
    wa, wb = model.random_init(12, 128, seed=net_seeds[0]), model.random_init(12, 128, seed=net_seeds[1])
    m = arena.Match(wa, wb, visits, games=games, dtype=dtype, seed=selfplay.DEFAULT_SEED if seed is None else seed)
    flops = model.flops_per_eval(12, 128)
    peak = MFMA_PEAK_TFLOPS[dtype]

    def launch_rate(d, tm):   # one tower launch's own rate: evaluations per launch x FLOPs / its average duration
        it = max(tm["iterations"], 1)
        ms = tm["net_ms"] / it
        evals = d["nn_evals"] / float(max(d["iterations"], 1))
        return {"evals_per_launch": evals, "tower_ms_per_launch": ms,
                "per_launch_tflops": evals * flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}

    try:
        m.set_game_limit(games)   # a slot whose cohort game is over goes idle (uai_ringmaster.py does the same)
        m.run(5)            # first launches (weight packing, kernel load) outside the region
        m.engine.sync()
        st0 = m.engine.stats()
        m.engine.timing_reset(TIMING_STRIDE)
        t0 = time.perf_counter()
        done, wins, annulled, plies, rounds = 0, {"a": 0.0, "b": 0.0}, 0, 0, 1
        p99 = thick = None          # (wall s, iterations) when 99 % of the games were over; the phase with >= 512 games live
        lost = None
        m.run(ROUND)
        while done < games:
            m.fetch()
            # between the fetch and the next run the engine is idle: counters and event timings cost no waiting here
            if games > 512 and thick is None and games - done <= 512:
                st, tm = m.engine.stats(), m.engine.timing()
                thick = (time.perf_counter() - t0, ROUND * rounds, {k: st[k] - st0[k] for k in st}, tm)
            if rounds % 16 == 0:
                lost = m.lost_games()
            m.run(ROUND)    # the next iterations run while the finished games are parsed and scored
            rounds += 1
            for g in m.drain():
                if g["uid"] >= games:
                    continue   # a replacement game in a slot whose cohort game is over
                done += 1
                plies += len(g["moves"])
                white = g["white"]
                black = "b" if white == "a" else "a"
                if g["result"] in (1, 2):
                    wins[white if g["result"] == 1 else black] += 1
                else:
                    wins["a"] += 0.5
                    wins["b"] += 0.5
                    annulled += 1
            if p99 is None and done >= 0.99 * games:
                p99 = (time.perf_counter() - t0, ROUND * (rounds - 1))   # (the games of the iterations fetched this round)
            if lost is not None and done < games:
                raise RuntimeError("config5 leg: %s, but only %d of %d were handed out" % (lost, done, games))
        m.engine.sync()
        dt = time.perf_counter() - t0
        st1 = m.engine.stats()
        tm1 = m.engine.timing()
        d = {k: st1[k] - st0[k] for k in st1}
        d["iterations"] = ROUND * rounds
        tf = d["nn_evals"] * flops / dt / 1e12
        whole = launch_rate(d, tm1)
        out = {"workload": "uai_ringmaster.py: %d-game match (fixed cohort, every pairing both ways) of two random-init 12x128 "
                           "nets (seeds %d, %d), %d visits/move, %s, all games in flight from the start"
                           % (games, net_seeds[0], net_seeds[1], visits, dtype),
               "net_seeds": list(net_seeds),
               "games": done, "wall_s": dt, "games_per_s": done / dt, "mcts_steps_per_s": d["steps"] / dt,
               "nn_evals_per_s": d["nn_evals"] / dt, "plies_per_s": d["plies"] / dt, "mean_plies": plies / float(max(done, 1)),
               "search_iterations": ROUND * rounds, "ms_per_iteration": 1e3 * dt / (ROUND * rounds),
               "tower_tflops": tf, "tower_frac_of_peak": tf / peak,
               "per_launch_tflops": whole["per_launch_tflops"], "per_launch_frac": whole["per_launch_tflops"] / peak,
               "evals_per_launch": whole["evals_per_launch"], "tower_ms_per_launch": whole["tower_ms_per_launch"],
               "wall_s_until_99pct_games": p99[0] if p99 else None, "iterations_until_99pct": p99[1] if p99 else None,
               "two_nets_in_one_launch": os.environ.get("AZH_ARENA_PAIR", "1") != "0",
               "score": "%s - %s (annulled: %d)" % (wins["a"], wins["b"], annulled),
               "reference": "uai_ringmaster.py:75-160,221-265 with two `uai_interface.py --visits %d` subprocesses "
                            "(engine.py's Python MCTS, about 1.5 k steps/s by its author's constant; not runnable here: TensorFlow-1)" % visits}
        if thick is not None:
            t_a, it_a, d_a, tm_a = thick
            d_a["iterations"] = it_a
            a = launch_rate(d_a, tm_a)
            tf_a = d_a["nn_evals"] * flops / t_a / 1e12
            out.update({"wall_s_while_ge512_live": t_a, "iterations_while_ge512_live": it_a,
                        "steps_per_s_while_ge512_live": d_a["steps"] / t_a, "nn_evals_per_s_while_ge512_live": d_a["nn_evals"] / t_a,
                        "ms_per_iteration_while_ge512_live": 1e3 * t_a / it_a,
                        "tower_frac_of_peak_while_ge512_live": tf_a / peak,
                        "per_launch_frac_while_ge512_live": a["per_launch_tflops"] / peak,
                        "evals_per_launch_while_ge512_live": a["evals_per_launch"],
                        "tower_ms_per_launch_while_ge512_live": a["tower_ms_per_launch"]})
        return out
    finally:
        m.close()


def config1_leg(games=2000):
    """BASELINE configs[0] (`generate_games.py --random-play --game-count 2000`) through this repo's drop-in: uniformly
    random games by the HIP playout kernel, entries built and json-dumped on the host exactly as generate_games.py writes
    them (to /dev/null here).  Reported beside the reference script's own rate, measured in the build container
    (26.9 games/s, one core, pure Python)."""
    from ataxxzero_amd import selfplay
    selfplay.random_play_entries(64, 1)            # first launch of the kernel
    t0 = time.perf_counter()
    entries = selfplay.random_play_entries(games, 20260101)
    t1 = time.perf_counter()
    written = plies = 0
    with open(os.devnull, "w") as sink:
        for e in entries:
            if e["result"] is not None:
                sink.write(json.dumps(e) + "\n")
                written += 1
                plies += len(e["moves"])
    t2 = time.perf_counter()
    return {"workload": "generate_games.py --random-play --game-count %d (no-blocker start, 400-ply cap)" % games,
            "games": written, "games_per_s": written / (t2 - t0), "plies_per_s": plies / (t2 - t0),
            "playout_kernel_and_copy_s": t1 - t0, "host_entries_and_json_s": t2 - t1,
            "mean_plies": plies / float(max(written, 1)),
            "reference_script_games_per_s": 26.9, "reference_source": "BASELINE.md §2 (unmodified script, 1 core, this repo's build container)"}


def spawn_ranks(n):
    """--gpus N with no launcher: one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its
    environment, exactly what torchrun would set), started before this process has made any HIP or torch call.
    Rank 0's stdout is passed through; any failing rank fails the run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    # a rank that dies would leave the others waiting at a barrier for ever: watch them all, and take the job down
    # with the first failure
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %s" % ", ".join("rank %d -> exit %d" % rc for rc in bad))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps (a step = --iters-per-step search iterations + one drain)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--iters-per-step", type=int, default=ITERS_PER_STEP)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--streams", type=int, default=2,
                    help="half-batches in flight per GPU: the games are split into this many engines, each with its own HIP "
                         "streams, sharing one set of packed weights (the reference's double buffer, cpp/self_play_client.cpp:"
                         "593-600: B of its 2B games are evaluated at a time).  2: one half's tower fills the ramp, the drain "
                         "and the tree phase of the other's (+10 %% at 4096 games, profiles/round4_half_batches_and_reserved_cus.txt)")
    ap.add_argument("--phase-fill", type=int, default=1000,
                    help="untimed set-up: the slots are loaded with steady-state positions (profiles/"
                         "round2_steady_state_positions.npz) and the trees are grown for this many iterations at full "
                         "sims/move before the --warmup steps; 0 = cold start (every game at ply 0)")
    ap.add_argument("--select-budget", type=int, default=48,
                    help="tree levels per select launch and game (azh_config.select_budget; 0 = unlimited): deeper "
                         "descents park and resume next iteration, so a launch does not last as long as the deepest "
                         "line of the batch; every game still plays exactly the same search")
    ap.add_argument("--eval-cache", action="store_true",
                    help="AZH_FLAG_EVAL_CACHE: positions a game's search has already evaluated are not sent to the net "
                         "again (engine.py's NNEvaluator.cache).  Off in the headline: the C++ generator evaluates every "
                         "new node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-ceiling", action="store_true",
                    help="skip the vendor-library bf16 GEMM measured beside the roofline (context for roofline.frac)")
    ap.add_argument("--no-target-leg", action="store_true", help="skip the 16384-game leg reported beside the headline")
    ap.add_argument("--legs", default="all",
                    help="comma-separated legs to measure beside the headline (default all): one_batch, target_10k_games, "
                         "target_10k_games_two_half_batches, "
                         "with_eval_cache, target_10k_games_with_eval_cache, with_f16, config1_random_play, config2, config4, "
                         "config5_arena, config5_arena_balanced")
    ap.add_argument("--other-configs-games", type=int, default=0,
                    help="games of the config2 / config4 / config5_arena legs (default: BASELINE's 4096 / 4096 / 1000)")
    ap.add_argument("--arena-ab", action="store_true",
                    help="also play the config5_arena leg with the two nets' towers as two launches back to back (AZH_ARENA_PAIR=0)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--plumbing-selftest", action="store_true",
                    help="NO GPU WORK: every rank reports fixed units so the launch / barrier / aggregation path of "
                         "--gpus N can be exercised on a CPU-only box (tests/test_distrib_gloo.py); value is null")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    from ataxxzero_amd import distrib
    # every rank's host threads (its own loop, the HIP runtime's helpers) on a core share of its own, taken before any
    # of them exists: the host side is all the ranks of a node have in common
    _, local_rank, _ = distrib.world()
    cores = distrib.pin_host_threads(local_rank, distrib.local_world())
    group = distrib.Group()
    if group.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, group.world))

    if args.plumbing_selftest:
        if os.environ.get("AZH_SELFTEST_FAIL_RANK") == str(group.rank):
            raise SystemExit(3)   # rehearse a rank dying before the barrier (tests/test_distrib_gloo.py)
        group.barrier()
        units, secs = 1000.0 * (group.rank + 1), 1.0 + group.rank
        total, t_max, rate = distrib.aggregate(group, units, secs)
        # (AZH_SELFTEST_SAME_CARD: every rank reports one bus id, to rehearse how two ranks on one card show up)
        bus = "0000:%02x:00.0" % (5 if os.environ.get("AZH_SELFTEST_SAME_CARD") else 5 + group.local_rank)
        per_rank = distrib.per_rank_report(group, units / secs, 1e3 * secs, distrib.device_for(group.local_rank, 0), bus, cores)
        group.barrier()
        if group.rank == 0:
            print(json.dumps({"metric": "plumbing selftest (no GPU work)", "value": None, "n_gpus": group.world,
                              "data": "none", "units_total": total, "t_max": t_max, "rate": rate, "per_rank": per_rank,
                              "seeds": [distrib.shard_seed(20260101, r) for r in range(group.world)]}))
        group.close()
        return 0

    from ataxxzero_amd import link, model, selfplay
    ndev = link.require_gpu()
    device = distrib.device_for(group.local_rank, ndev)
    link.check(link.load().azh_set_device(device))

    conv, bn = model.random_init(args.blocks, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=args.games, visits=args.visits, dtype=args.dtype,
                           seed=distrib.shard_seed(selfplay.DEFAULT_SEED, group.rank), streams=args.streams,
                           select_budget=args.select_budget, flags=link.FLAG_EVAL_CACHE if args.eval_cache else 0)
    ages = spread(sp, args, distrib.shard_seed(selfplay.DEFAULT_SEED, group.rank))
    d, finished, dt, tm, finished_per_step = measure(sp, args, args.steps, args.warmup, group)
    steps_total, t_max, rate = distrib.aggregate(group, d["steps"], dt)
    evals_total = group.reduce(d["nn_evals"], "sum")
    plies_total = group.reduce(d["plies"], "sum")
    games_total = group.reduce(d["games"], "sum")     # finished games (result 1 or 2), device counter
    written_total = group.reduce(finished, "sum")      # of which written out (games from loaded positions are not)
    # each rank's own figures, device and cores beside the aggregate (a straggler, or two ranks on one card, must show)
    per_rank = distrib.per_rank_report(group, d["steps"] / dt, 1e3 * dt / args.steps, device, link.pci_bus_id(device), cores)

    if group.rank == 0:
        flops = model.flops_per_eval(args.blocks, 128)
        iters = args.steps * args.iters_per_step
        it = max(tm["iterations"], 1)                # sampled launches (per half-batch when streams > 1)
        launch_ms = tm["net_ms"] / it
        evals_per_launch = d["nn_evals"] / float(iters * args.streams)
        launch_tf = evals_per_launch * flops / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
        # ONE definition whatever the number of half-batches: `achieved` is the kernel's rate over the chip — every FLOP of
        # the kernel in the timed region / the region's wall time (rank 0's) — a LOWER bound of the rate while a tower runs
        # (the region also holds the moments in which none does: tree phases with one batch, nothing much with two).  With
        # half-batches in flight two launches share the chip most of the time and a launch's own duration prices about half
        # a chip; one launch's own rate is reported as `per_launch` (and, every launch alone on the chip, by the one_batch leg).
        region_tf = d["nn_evals"] * flops / dt / 1e12
        achieved_tf = region_tf
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        traffic, traffic_src = measured_traffic(args.streams)
        tree_ms = (tm["select_ms"] + tm["backup_ms"]) / it
        tree_gbs = tree_bytes(d) / float(iters * args.streams) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
        out = {
            "metric": "MCTS node-evals/s at %d sims/move (self-play; games/s beside it)" % args.visits,
            "value": rate, "unit": "node-evals/s", "n_gpus": group.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init .npy-layout weights, seed 1; "
                                                              "self-play from the reference start position)",
            "config": {"workload": "%d concurrent self-play games per GPU%s, %d sims/move, %dx128 conv net, %s "
                                   "(per-GPU shard of BASELINE configs[2]; configs[1] at the metric's 400 sims)"
                                   % (args.games, "" if args.streams == 1 else " in %d half-batches of %d, each with its own streams"
                                      % (args.streams, args.games // args.streams), args.visits, args.blocks, args.dtype),
                       "step": "%d search iterations over the whole batch + one drain of the finished games"
                               % args.iters_per_step,
                       "games_per_gpu": args.games, "half_batches_in_flight": args.streams, "visits": args.visits,
                       "select_budget": args.select_budget, "eval_cache": bool(args.eval_cache),
                       "setup": ("slots loaded with steady-state positions built from one complete generation of real games of "
                                 "this workload (%s, mean ply %.0f), trees grown for %d untimed iterations, then the warm-up"
                                 % (ages["source"], ages["mean_ply"], args.phase_fill)) if ages else "cold start",
                       "net": "%dx128" % args.blocks,
                       "parallelism": "%d independent game shards, no collective" % group.world,
                       "timing_barrier_backend": group.backend},
            "ms_per_iteration": 1e3 * t_max / iters,
            "nn_evals_per_s": evals_total / t_max, "plies_per_s": plies_total / t_max,
            "games_per_s": games_total / t_max,    # games finished (result 1 or 2) in the timed region, counted on the device
            "game_lines_written_in_timed_region": written_total,
            "games_finished_in_timed_region": games_total,
            "games_finished_per_step_rank0": finished_per_step,   # flat = the loaded state is the steady state
            "mean_plies_per_finished_game": plies_total / games_total if games_total else None,
            # the counted figure rests on a few hundred games finishing inside the region (+-5 %); every ply played in it is a
            # 1 / (mean game length) share of a game: plies/s over the mean length of the generation the loaded state was drawn from
            "games_per_s_from_plies": (plies_total / t_max / SNAPSHOT_MEAN_GAME_PLIES) if ages and args.visits == 400 else None,
            "games_per_s_from_plies_is": "plies_per_s / %.2f plies per game (profiles/round2_steady_state_positions.npz, generation 0)" % SNAPSHOT_MEAN_GAME_PLIES,
            "roofline": {"bound": "mfma", "kernel": "%s<%s>" % ("k_tower" if args.dtype == "f32" or os.environ.get("AZH_TOWER") == "1" else "k_tower2", args.dtype), "achieved": achieved_tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved_tf / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_is": "the kernel's rate over the chip: all its FLOPs in the timed region / the region's wall time "
                                        "(%d launch(es) share the chip: per_launch is one of them; one_batch leg = a launch alone)" % args.streams,
                         "kernel_flops_in_region": d["nn_evals"] * flops, "region_s": dt,
                         "per_launch": {"evals_per_launch": evals_per_launch, "avg_launch_ms": launch_ms, "achieved": launch_tf,
                                        "frac": launch_tf / peak, "launches_sharing_the_chip": args.streams},
                         "per_launch_frac": launch_tf / peak,
                         "avg_launch_ms": launch_ms, "evals_per_launch": evals_per_launch,
                         "launches_timed": it, "launches_in_region": iters * args.streams, "flops_per_eval": flops},
            "tree_roofline": {"bound": "hbm", "kernels": "k_tree (backup + move-due mark + select + leaf-list compaction, one launch, four games per "
                                         "workgroup, one beyond 8192 games) on the engine's stream; the queued moves (re-roots) are played by "
                                         "the first workgroups of the tower launch",
                              "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tree_gbs / HBM_PEAK_GBS,
                              # what this chip delivers for the launch's own access pattern — dependent scattered 672-byte
                              # reads, one chain per wave (tools/microbench/random_chase.hip): 3.1 TB/s with 4096 chains in
                              # flight, 5.3 TB/s from 8192 on; `peak` stays the guide's streaming figure
                              "scattered_read_roofline": {"value": SCATTERED_PEAK_GBS, "unit": "GB/s", "at_chains_in_flight": ">= 8192",
                                                          "frac": tree_gbs / SCATTERED_PEAK_GBS,
                                                          "source": "profiles/round3_random_chase.txt"},
                              "tree_phase_ms_per_iteration": tree_ms,
                              "tree_phase_is": ("tower end -> next tower start on the engine's stream" if args.streams == 1 else
                                                "tower end -> next tower start of ONE half-batch: its tree launch waits for the slots the "
                                                "other half's tower workgroups free, and runs under that tower (one_batch has the launch alone)"),
                              "bytes_per_step": tree_bytes(d) / float(max(d["steps"], 1)),
                              "levels_per_step": d["levels"] / float(max(d["steps"], 1)),
                              "children_per_step": d["children"] / float(max(d["steps"], 1))},
            "per_rank": per_rank,
            "counters": d,
        }
        sp.close()
        def want(leg):
            return group.world == 1 and not args.no_target_leg and (args.legs == "all" or leg in args.legs.split(","))

        if want("one_batch") and args.streams != 1:
            # the headline workload as ONE batch: the tower launch and the tree launch alone on the chip, i.e. the
            # kernels' own rooflines (frac = FLOPs per launch / launch duration; tree bytes / tree phase)
            ob = out["one_batch"] = target_leg(conv, bn, args, games=args.games, steps=6, warmup=2)
            alone = "the one_batch leg of this line: the same workload as one batch, every launch alone on the chip"
            out["roofline"]["launch_alone_on_the_chip"] = {
                "evals_per_launch": ob["evals_per_launch"], "avg_launch_ms": ob["tower_ms_per_launch"],
                "achieved": ob["per_launch_tflops"], "frac": ob["per_launch_frac"], "source": alone}
            out["tree_roofline"]["launch_alone_on_the_chip"] = {
                "tree_phase_ms_per_iteration": ob["tree_ms_per_iteration"], "frac": ob["tree_roofline_frac"],
                "achieved": ob["tree_roofline_frac"] * HBM_PEAK_GBS, "source": alone}
        if want("target_10k_games"):
            out["target_10k_games"] = target_leg(conv, bn, args)
        if want("target_10k_games_two_half_batches") and args.streams != 1:
            # the same 16384 games as two half-batches of 8192 in flight (+1.8 % with round 5's tree kernels; the one-batch leg
            # above keeps the launches alone on the chip: its fractions are the kernels' own)
            out["target_10k_games_two_half_batches"] = target_leg(conv, bn, args, streams=args.streams)
        if not args.eval_cache:
            # the same workload with AZH_FLAG_EVAL_CACHE (the generator CLI's default): MCTS steps/s and games/s rise,
            # net evaluations/s do not (the headline keeps the C++ generator's rule: every new node goes to the net)
            if want("with_eval_cache"):
                out["with_eval_cache"] = target_leg(conv, bn, args, games=args.games, steps=6, warmup=2,
                                                    flags=link.FLAG_EVAL_CACHE, streams=args.streams)
            if want("target_10k_games_with_eval_cache"):
                out["target_10k_games_with_eval_cache"] = target_leg(conv, bn, args, flags=link.FLAG_EVAL_CACHE,
                                                                     select_budget=64)
        if args.dtype == "bf16" and want("with_f16"):
            # the f16 tower on the headline workload: closer to the f32 search than bf16 (profiles/
            # round2_precision_in_the_loop.json: top-1 100 % / TV 0.02 % vs 98.4 % / 1.7 %); BASELINE names bf16
            out["with_f16"] = target_leg(conv, bn, args, games=args.games, steps=6, warmup=2, dtype="f16", streams=args.streams)
        if want("config1_random_play"):
            out["config1_random_play"] = config1_leg()
        # BASELINE.json's other configs, measured like the headline (6 steps each) — beside the default headline, or when
        # asked for by name (--legs)
        default_headline = (args.visits, args.blocks, args.dtype, args.games) == (400, 12, "bf16", 4096)
        og = args.other_configs_games
        if want("config2") and (default_headline or args.legs != "all"):
            # configs[1]: 4096 games, 200 sims/move, the 12x128 net, bf16
            c2 = model.random_init(12, 128, seed=1)
            out["config2"] = target_leg(c2[0], c2[1], args, games=og or 4096, steps=6, warmup=2, visits=200, blocks=12,
                                        dtype="bf16", streams=args.streams)
        if want("config4") and (default_headline or args.legs != "all"):
            # configs[3]: 8 blocks x 128, fp16 MFMA, 800 sims/move (random-init seed 3)
            c4 = model.random_init(8, 128, seed=3)
            out["config4"] = target_leg(c4[0], c4[1], args, games=og or 4096, steps=6, warmup=2, visits=800, blocks=8,
                                        dtype="f16", streams=args.streams)
        if want("config5_arena") and (default_headline or args.legs != "all"):
            # configs[4]: 1000-game arena of two nets
            out["config5_arena"] = config5_leg(games=og or 1000)
        if want("config5_arena_balanced") and (default_headline or args.legs != "all"):
            # the same match between two nets that are a match for each other (seeds 27 : 28: 51 - 49 over a 100-game
            # pre-match, no game cut, profiles/round6_arena_pairing_search.txt): seed 1 loses 99 % of its games to seed 2
            # inside 90 plies, and the leg above then mostly prices the one game that is cut at 400 plies
            out["config5_arena_balanced"] = config5_leg(games=og or 1000, net_seeds=(27, 28))
        if want("config5_arena") and (default_headline or args.legs != "all"):
            if args.arena_ab:
                # the same match with the two nets' towers launched one after the other (the round-4 loop), same box, same call
                os.environ["AZH_ARENA_PAIR"] = "0"
                try:
                    out["config5_arena_two_launches"] = config5_leg(games=og or 1000)
                finally:
                    del os.environ["AZH_ARENA_PAIR"]
        if group.world == 1 and not args.no_gemm_ceiling:
            out["roofline"]["vendor_gemm_on_this_box"] = vendor_gemm_ceiling()
        if group.world == 1 and not args.no_cpu_baseline:
            net = link.Net(conv, bn)
            out["cpu_baseline"] = cpu_baseline(net, args.visits, args.dtype, args.cpu_seconds)
            net.close()
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        sp.close()
    group.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
