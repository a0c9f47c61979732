#!/usr/bin/env python3
"""Length distribution of one complete GENERATION of self-play games at the bench's workload (4096 games, 400 sims/move,
12x128 random-init seed-1 net, bf16): the real CLI runs with uid-ordered emission until every game of generation 0
(uids 0..4095) has been written, so the first 4096 lines are an unbiased sample of game lengths — short and long games
alike.  bench.py's set-up draws the slots' ages from the stationary age distribution this implies.

    python tools/length_histogram.py [--games 4096] [--visits 400] [--max-seconds 260] > profiles/roundN_game_lengths.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ataxxzero_amd import model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--max-seconds", type=float, default=260.0)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        net = os.path.join(tmp, "model-001.npy")
        conv, bn = model.random_init(args.blocks, 128, seed=1)
        model.save_model(net, conv, bn)
        out = os.path.join(tmp, "model-001-0.json")
        t0 = time.time()
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net,
                                 "--output-games", out, "--visits", str(args.visits), "--buffer-size", str(args.games // 2),
                                 "--seed", "20260101", "--emit-order", "uid", "--max-seconds", str(args.max_seconds)],
                                cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        log = proc.communicate()[0].decode()
        wall = time.time() - t0
        lengths, results = [], []
        with open(out) as f:
            for line in f:
                if line.strip():
                    e = json.loads(line)
                    lengths.append(len(e["moves"]))
                    results.append(e["result"])
    rate = [l for l in log.splitlines() if l.startswith("Rate:")]
    gen0 = lengths[:args.games]
    hist = [0] * 401
    for n in gen0:
        hist[n] += 1
    print(json.dumps({"workload": "%d games, %d sims/move, %dx128 bf16" % (args.games, args.visits, args.blocks),
                      "wall_seconds": wall, "lines_written": len(lengths), "generation0_written": len(gen0),
                      "generation0_complete": len(lengths) >= args.games,
                      "mean_plies": sum(gen0) / max(len(gen0), 1), "min": min(gen0), "max": max(gen0),
                      "results_x_o": [results[:args.games].count(1), results[:args.games].count(2)],
                      "histogram_plies": hist, "last_rate_line": rate[-1] if rate else None}))


if __name__ == "__main__":
    main()
