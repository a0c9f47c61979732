#!/usr/bin/env python3
"""Steady-state positions for bench.py's set-up, built from complete real games by the renewal-process recipe.

A generator that has run for a long time finds a slot inside game i with probability proportional to the game's
length L_i, at a ply uniform on 0..L_i-1.  So: the real CLI plays one complete GENERATION at the bench's workload (4096
games, 400 sims/move, 12x128 random-init seed-1 net, bf16; uid-ordered emission makes the first 4096 lines exactly
generation 0, long and short games alike), and for every slot one of those games is drawn length-biased and one of
its plies uniformly; the position before that ply's move (the record's boards[ply]) and the ply are written out.
(Waiting for the steady state to form by itself does not work in bounded time: 300 s after a cold start the finish
rate still swings between 50 and 105 games/s with the generation period — tools/steady_state_snapshot.py.)

    python tools/steady_state_positions.py --out profiles/round2_steady_state_positions.npz
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ataxxzero_amd import model  # noqa: E402


def cells_to_bitboards(cells):
    """49 ints (index x + 7 y, y = 0 at rank 7) -> (x stones, o stones) with square = file + 7 * rank0."""
    x = o = 0
    for idx, v in enumerate(cells):
        if v:
            sq = (idx % 7) + 7 * (6 - idx // 7)
            if v == 1:
                x |= 1 << sq
            else:
                o |= 1 << sq
    return x, o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--max-seconds", type=float, default=340.0)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        net = os.path.join(tmp, "model-001.npy")
        conv, bn = model.random_init(args.blocks, 128, seed=1)
        model.save_model(net, conv, bn)
        out = os.path.join(tmp, "model-001-0.json")
        open(out, "w").close()
        t0 = time.time()
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net,
                                 "--output-games", out, "--visits", str(args.visits), "--buffer-size", str(args.games // 2),
                                 "--seed", str(args.seed), "--emit-order", "uid", "--max-seconds", str(args.max_seconds)],
                                cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        # stop as soon as generation 0 is complete (looper.py's own way: count the lines, SIGTERM)
        while proc.poll() is None:
            time.sleep(5.0)
            with open(out) as f:
                n = sum(1 for line in f if line.strip())
            print("t=%.0f s lines=%d" % (time.time() - t0, n), file=sys.stderr)
            if n >= args.games:
                proc.terminate()
                break
        proc.wait()
        games = []
        with open(out) as f:
            for line in f:
                if line.strip():
                    games.append(json.loads(line))
                    if len(games) == args.games:
                        break
    if len(games) < args.games:
        raise SystemExit("generation 0 incomplete: %d of %d games" % (len(games), args.games))
    lengths = np.array([len(g["moves"]) for g in games], dtype=np.float64)
    rng = np.random.default_rng(args.seed)
    pick = rng.choice(len(games), size=args.games, p=lengths / lengths.sum())
    boards = np.zeros((args.games, 2), dtype=np.uint64)
    plies = np.zeros(args.games, dtype=np.int32)
    for slot, gi in enumerate(pick):
        ply = int(rng.integers(0, int(lengths[gi])))
        x, o = cells_to_bitboards(games[gi]["boards"][ply])
        boards[slot] = (x | ((ply & 1) << 63), o)      # x moves on even plies (train.py:50)
        plies[slot] = ply
    meta = {"workload": "%d games, %d sims/move, %dx128 bf16 (seed-1 random-init net)" % (args.games, args.visits, args.blocks),
            "recipe": "game drawn with probability ~ length among the %d games of generation 0, ply uniform within it" % len(games),
            "generation0_mean_plies": float(lengths.mean()), "mean_ply_of_sample": float(plies.mean()),
            "expected_mean_age": float((lengths ** 2).sum() / lengths.sum() / 2 - 0.5), "wall_seconds": time.time() - t0}
    np.savez_compressed(args.out, boards=boards, plies=plies, meta=np.asarray(json.dumps(meta)))
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
