#!/usr/bin/python
"""Does the closed loop learn?  Generation -> training -> generation with the new net, then every generation against
the first one in the arena.

This script issues the three commands the reference's orchestration issues — the generator and the trainer exactly as
looper.py:33-41 and looper.py:140-148 spell them (plus the extension flags named below), the arena as
uai_ringmaster.py is used with two `uai_interface.py --network-path X --visits V` engines — against this repo's
drop-in scripts, on one GPU, and writes a summary: the score of model-00k against model-001, and per generation the
generator's own counters (dropped / edge_overflow / ring_overflow / parked share / re-root spills / cache hits) and its
rate with the TRAINED net in the search (a trained net searches shallower and keeps larger subtrees than a random one).

It is the one check of generator <-> trainer <-> arena conventions together: value sign, side to move from ply
parity, the dists -> heat-map mapping under the 8 symmetries, batch-norm statistics in the .npy,
blockers in self-play but not in training (Q2).  A mistake in any of them shows up as a net that does not improve.

    python tools/learning_curve.py --prefix /tmp/run1 --iterations 5 --game-count 2000 --visits 400 \\
        --training-steps 1000 --arena-games 1000 --arena-visits 100 --out gpurun_out/learning_curve.txt
"""
import argparse
import json
import os
import re
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model_path(prefix, i):
    return os.path.join(prefix, "models", "model-%03i.npy" % i)      # looper.py:67-68


def games_path(prefix, i, process_index=0):
    return os.path.join(prefix, "games", "model-%03i-%i.json" % (i, process_index))   # looper.py:70-74


def count_games(path):
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        return sum(1 for line in f if line.strip())


def log(out, text):
    print(text)
    sys.stdout.flush()
    with open(out, "a") as f:
        f.write(text + "\n")


def generate(args, n, out):
    """looper.generate_games (looper.py:22-65): start the generator, poll the line count, SIGTERM at --game-count."""
    path = games_path(args.prefix, n)
    open(path, "a").close()
    if count_games(path) >= args.game_count:
        log(out, "generation %d: enough games already" % n)
        return None
    cmd = [sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", model_path(args.prefix, n),
           "--output-games", path, "--visits", str(args.visits)] + args.generator_flags
    gen_log = os.path.join(args.prefix, "generator-%03i.log" % n)
    t0 = time.time()
    with open(gen_log, "w") as lf:
        proc = subprocess.Popen(cmd, cwd=ROOT, stdout=lf, stderr=subprocess.STDOUT)
        try:
            while proc.poll() is None:
                time.sleep(args.poll_seconds)
                have = count_games(path)
                print("  generation %d: %d games after %.0f s" % (n, have, time.time() - t0))
                sys.stdout.flush()
                if have >= args.game_count:
                    break
            if proc.poll() is None:
                proc.send_signal(signal.SIGTERM)
            rc = proc.wait(timeout=60)
        finally:
            if proc.poll() is None:
                proc.kill()
    seconds = time.time() - t0
    text = open(gen_log).read()
    totals = None
    for line in text.splitlines():
        if line.startswith("Totals: "):
            totals = json.loads(line[len("Totals: "):])
    if rc != 0 or totals is None:
        log(out, "generation %d: generator exit code %s\n%s" % (n, rc, text[-3000:]))
        raise SystemExit(2)
    rates = [float(m.group(1)) for m in re.finditer(r"^Rate: ([0-9.]+)k evals/s", text, re.M)]
    totals["wall_seconds"] = round(seconds, 1)
    totals["rate_lines_k_evals_per_s"] = rates
    return totals


def train(args, n, out):
    """looper.py:135-149: the window of game files and `train.py --steps S --games ... --old-path A --new-path B`."""
    low = min(n, max(args.training_window_exclude + 1, n - args.training_window + 1))
    paths = [games_path(args.prefix, i) for i in range(low, n + 1)]
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--steps", str(args.training_steps), "--games"] + paths + [
        "--old-path", model_path(args.prefix, n), "--new-path", model_path(args.prefix, n + 1)]
    t0 = time.time()
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    if res.returncode != 0:
        log(out, "training %d -> %d failed:\n%s\n%s" % (n, n + 1, res.stdout[-3000:], res.stderr[-3000:]))
        raise SystemExit(2)
    losses = re.findall(r"^Step: +(\d+) -- loss: ([0-9.]+) +\(policy: ([0-9.]+) +value: ([0-9.]+)\)", res.stdout, re.M)
    return {"games_files": [os.path.basename(p) for p in paths], "seconds": round(time.time() - t0, 1),
            "val_loss_first": [float(x) for x in losses[0][1:]] if losses else None,
            "val_loss_last": [float(x) for x in losses[-1][1:]] if losses else None}


def arena(args, k, out, other=1):
    """model-00k against model-00`other`, every pairing both ways, through the ringmaster drop-in."""
    eng = "python uai_interface.py --network-path %s --visits %d"
    cmd = [sys.executable, os.path.join(ROOT, "uai_ringmaster.py"),
           "--engine", eng % (model_path(args.prefix, k), args.arena_visits),
           "--engine", eng % (model_path(args.prefix, other), args.arena_visits),
           "--game-count", str(args.arena_games), "--seed", str(args.seed + 31 * k + other)]
    if args.arena_opening_depth > 0:
        cmd += ["--opening-depth", str(args.arena_opening_depth)]
    t0 = time.time()
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    if res.returncode != 0:
        log(out, "arena %d failed:\n%s\n%s" % (k, res.stdout[-3000:], res.stderr[-3000:]))
        raise SystemExit(2)
    wins = re.findall(r"^Wins: ([0-9.]+) - ([0-9.]+) \(annulled: (\d+)\)", res.stdout, re.M)
    a, b, ann = float(wins[-1][0]), float(wins[-1][1]), int(wins[-1][2])
    # an annulled game (cut at 400 plies: the rules have no repetition or move-count draw) counts half a point each in the
    # ringmaster's tally; over the DECISIVE games alone the score says who wins when somebody does
    decisive = a + b - ann
    return {"new": a, "first": b, "annulled": ann, "score": a / (a + b), "decisive_games": int(decisive),
            "decisive_score": (a - 0.5 * ann) / decisive if decisive > 0 else float("nan"),
            "seconds": round(time.time() - t0, 1)}


def match_line(label, r):
    return "%s: %.1f - %.1f (annulled %d) = %.1f %%; decisive games %d: %.1f %%   [%.0f s]" % (
        label, r["new"], r["first"], r["annulled"], 100.0 * r["score"], r["decisive_games"], 100.0 * r["decisive_score"], r["seconds"])


def value_check(model_file, games_file, samples=4096, seed=7):
    """Does the net's value head point the right way, and does its policy agree with the search it was trained on?  On
    positions of `games_file` drawn through the training pipeline's own encoders (training.make_minibatch: side to move from
    ply parity, value target +1 when the mover went on to win): correlation of the net's value with that target, the share
    of positions where the value has the target's sign, and the share where the policy's best move is the search's most
    visited move.  Evaluated by the HIP tower (f32) — the net exactly as the generator and the arena run it."""
    import random
    import numpy as np
    from ataxxzero_amd import link, model, training
    link.require_gpu()
    random.seed(seed)
    entries = training.load_entries([games_file])
    feats, pols, vals = training.make_minibatch(entries, samples)
    # feature rows [x][y][c] -> leaf boards (mover, opponent): bit x + 7 * (6 - y) (cpp/self_play_client.cpp:181)
    sq = np.array([[x + 7 * (6 - y) for y in range(7)] for x in range(7)], dtype=np.uint64)
    mover = (feats[..., 1].astype(np.uint64) << sq).sum(axis=(1, 2), dtype=np.uint64)
    opp = (feats[..., 2].astype(np.uint64) << sq).sum(axis=(1, 2), dtype=np.uint64)
    net = link.Net(*model.load_model(model_file))
    logits, values = net.forward(np.stack([mover, opp], axis=1), 0, link.DTYPE_F32)
    v, t = values.reshape(-1).astype(np.float64), vals.reshape(-1).astype(np.float64)
    legal = pols.reshape(samples, -1) > 0           # the search's visited moves (all legal moves it expanded)
    masked = np.where(legal, logits.reshape(samples, -1), -np.inf)
    print(json.dumps({"positions": samples, "value_target_correlation": float(np.corrcoef(v, t)[0, 1]),
                      "value_sign_agreement": float(np.mean(np.sign(v) == t)), "mean_abs_value": float(np.abs(v).mean()),
                      "policy_top1_is_most_visited": float(np.mean(masked.argmax(axis=1) == pols.reshape(samples, -1).argmax(axis=1)))}))


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--value-check":
        return value_check(sys.argv[2], sys.argv[3])
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--prefix", required=True, help="run directory (models/ and games/ are created in it)")
    ap.add_argument("--out", required=True, help="summary text file (appended to as the run proceeds)")
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--visits", type=int, default=400)                      # looper.py:107
    ap.add_argument("--game-count", type=int, default=2000)
    ap.add_argument("--training-steps", type=int, default=1000)
    ap.add_argument("--training-window", type=int, default=10)              # looper.py:111
    ap.add_argument("--training-window-exclude", type=int, default=3)       # looper.py:112
    ap.add_argument("--arena-games", type=int, default=1000)
    ap.add_argument("--arena-visits", type=int, default=100)
    ap.add_argument("--arena-opening-depth", type=int, default=0,
                    help="matches start from random openings of this many plies (uai_ringmaster.py --opening-depth): two "
                         "near-deterministic nets otherwise shuffle to the 400-ply cut game after game")
    ap.add_argument("--anchor", type=int, default=0, help="> 1: also play every later generation against model-<anchor>")
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--poll-seconds", type=float, default=10.0)             # looper.py:54
    ap.add_argument("--precision-positions", type=int, default=0,
                    help="> 0: finish with tools/precision_in_the_loop.py on the last net (bf16 / f16 search against the f32 "
                         "search on that many positions)")
    ap.add_argument("--generation-deadline", type=float, default=0.0,
                    help="> 0: start no further generation after this many seconds (a GPU box call is bounded); the matches "
                         "and checks then cover the generations that were completed")
    ap.add_argument("--generator-flags", type=str, default="--buffer-size 1024",
                    help="extension flags appended to the generator command (one string)")
    args = ap.parse_args()
    args.generator_flags = args.generator_flags.split()
    from ataxxzero_amd import model

    os.makedirs(os.path.join(args.prefix, "models"), exist_ok=True)
    os.makedirs(os.path.join(args.prefix, "games"), exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    out = args.out
    log(out, "# learning curve: %s" % json.dumps({k: v for k, v in vars(args).items()}, sort_keys=True))
    if not os.path.exists(model_path(args.prefix, 1)):
        conv, bn = model.random_init(args.blocks, args.filters, seed=args.seed)
        model.save_model(model_path(args.prefix, 1), conv, bn)
        log(out, "model-001: random init (model.py:103-114 distributions), seed %d" % args.seed)

    t_start = time.time()
    for n in range(1, args.iterations + 1):
        if args.generation_deadline > 0 and time.time() - t_start > args.generation_deadline:
            log(out, "deadline of %.0f s passed after %d generations: going on to the matches" % (args.generation_deadline, n - 1))
            args.iterations = n - 1
            break
        if not os.path.exists(model_path(args.prefix, n + 1)):
            g = generate(args, n, out)
            if g is not None:
                steps_per_s = g["steps"] / g["seconds"]
                iters = g["steps"] + g["parked"]  # (game, iteration) pairs that searched: a step, or a parked descent
                log(out, "generation %d (model-%03i plays): %d games written in %.0f s; %.3f M node-evals/s, %.3f M "
                         "net evals/s, %.1f plies/game; dropped %d, edge_overflow %d, ring_overflow %d, parked share "
                         "%.2f %%, re-roots %d (spilled %d, kept nodes/re-root %.1f), cache hits %.1f %% of expansions; "
                         "levels/step %.2f, children/level %.1f" % (
                             n, n, g["written"], g["wall_seconds"], steps_per_s * 1e-6, g["nn_evals"] / g["seconds"] * 1e-6,
                             g["plies"] / max(1, g["games"] + g["dropped"]), g["dropped"], g["edge_overflow"],
                             g["ring_overflow"], 100.0 * g["parked"] / max(1, iters), g["plies"], g["reroot_spills"],
                             g["reroot_nodes"] / max(1, g["plies"]),
                             100.0 * g["cache_hits"] / max(1, g["cache_hits"] + g["nn_evals"] - g["plies"]),
                             g["levels"] / max(1, g["steps"]), g["children"] / max(1, g["levels"])))
            t = train(args, n, out)
            log(out, "training %d -> %d: %s, %.0f s; validation loss (total, policy, value) %s -> %s" % (
                n, n + 1, " ".join(t["games_files"]), t["seconds"], t["val_loss_first"], t["val_loss_last"]))
        else:
            log(out, "model-%03i exists, skipping iteration %d" % (n + 1, n))
    log(out, "")
    log(out, "arena: model-00k vs model-001, %d games (every pairing both ways), %d visits/move" % (
        args.arena_games, args.arena_visits))
    if args.arena_opening_depth > 0:
        log(out, "(every pairing from its own opening of %d uniformly random plies, played both ways: uai_ringmaster.py:185-196)"
            % args.arena_opening_depth)
    for k in range(2, args.iterations + 2):
        log(out, match_line("model-%03i vs model-001" % k, arena(args, k, out)))
    log(out, "")
    log(out, "arena: every generation against its parent (the net it was trained from), same match size")
    for k in range(3, args.iterations + 2):
        log(out, match_line("model-%03i vs model-%03i" % (k, k - 1), arena(args, k, out, other=k - 1)))
    if args.anchor > 1 and args.anchor <= args.iterations + 1:
        log(out, "")
        log(out, "arena: the later generations against the anchor model-%03i (a net that already plays: the first net stops "
                 "telling generations apart once they all beat it every time)" % args.anchor)
        for k in range(args.anchor + 1, args.iterations + 2):
            log(out, match_line("model-%03i vs model-%03i" % (k, args.anchor), arena(args, k, out, other=args.anchor)))
    log(out, "")
    log(out, "value head and policy of model-00k on the games model-00k itself went on to play (positions it was NOT trained on), "
             "through training.make_minibatch's encoders and the HIP tower (f32):")
    for k in range(1, args.iterations + 1):
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--value-check", model_path(args.prefix, k),
                              games_path(args.prefix, k)], cwd=ROOT, capture_output=True, text=True)
        if res.returncode != 0:
            log(out, "value check %d failed:\n%s" % (k, res.stderr[-2000:]))
            raise SystemExit(2)
        log(out, "model-%03i on model-%03i-0.json: %s" % (k, k, res.stdout.strip().splitlines()[-1]))
    if args.precision_positions > 0:
        last = args.iterations + 1
        log(out, "")
        log(out, "reduced precision in the search of the TRAINED net model-%03i (tools/precision_in_the_loop.py: %d mid-game "
                 "positions, 400 sims, bf16 / f16 tower against the f32 tower, same seeds, no root noise):" % (last, args.precision_positions))
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "precision_in_the_loop.py"), "--positions",
                              str(args.precision_positions), "--network", model_path(args.prefix, last)], cwd=ROOT,
                             capture_output=True, text=True)
        log(out, res.stdout.strip().splitlines()[-1] if res.returncode == 0 else "failed:\n" + res.stderr[-2000:])


if __name__ == "__main__":
    main()
