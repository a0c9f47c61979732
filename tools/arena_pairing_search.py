#!/usr/bin/env python3
"""Which pair of random-init 12x128 nets makes a BALANCED config-5 match?

bench.py's config5 leg plays seed 1 against seed 2 (BASELINE's "two .npy nets"): net 1 is degenerate against net 2 —
7.5 : 992.5, mean game 85 plies, 99 % wipe-outs — so after a fifth of the match's iterations nearly every game is over and
the rest prices the latency of a near-empty batch.  This tool plays a short pre-match (100 games, the leg's settings: 100
visits, f16, all games in flight) for a list of seed pairs and prints score, mean plies and annulled games; the leg's second
pairing is picked from its output once and hard-coded, with this output kept under profiles/.

    python tools/arena_pairing_search.py [--pairs 3:4,5:6,...] [--games 100] [--visits 100]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ataxxzero_amd import arena, model, selfplay  # noqa: E402


def pre_match(seed_a, seed_b, games, visits, dtype):
    wa, wb = model.random_init(12, 128, seed=seed_a), model.random_init(12, 128, seed=seed_b)
    m = arena.Match(wa, wb, visits, games=games, dtype=dtype, seed=selfplay.DEFAULT_SEED)
    m.set_game_limit(games)
    wins, annulled, plies, done = {"a": 0.0, "b": 0.0}, 0, [], 0
    m.run(50)
    while done < games:
        m.fetch()
        m.run(50)
        for g in m.drain():
            done += 1
            plies.append(len(g["moves"]))
            white = g["white"]
            black = "b" if white == "a" else "a"
            if g["result"] in (1, 2):
                wins[white if g["result"] == 1 else black] += 1
            else:
                wins["a"] += 0.5
                wins["b"] += 0.5
                annulled += 1
    m.close()
    plies.sort()
    return wins["a"], wins["b"], annulled, sum(plies) / len(plies), plies[len(plies) // 2], plies[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="1:2,3:4,5:6,7:8,9:10,11:12,1:3,2:4,5:7,6:8")
    ap.add_argument("--games", type=int, default=100)
    ap.add_argument("--visits", type=int, default=100)
    ap.add_argument("--dtype", default="f16")
    args = ap.parse_args()
    selfplay.select_device(0)
    print("pre-matches of %d games, %d visits/move, %s, two random-init 12x128 nets (model.random_init seeds a : b)"
          % (args.games, args.visits, args.dtype))
    print("%-9s %-15s %-9s %-11s %-13s %s" % ("seeds", "score a - b", "annulled", "mean plies", "median plies", "longest"))
    for pair in args.pairs.split(","):
        a, b = (int(v) for v in pair.split(":"))
        wa, wb, ann, mean, med, longest = pre_match(a, b, args.games, args.visits, args.dtype)
        share = wa / float(wa + wb)
        print("%-9s %-15s %-9d %-11.1f %-13d %-8d %s" % ("%d : %d" % (a, b), "%g - %g" % (wa, wb), ann, mean, med, longest,
                                                        "balanced (%.0f %%)" % (100 * share) if 0.35 <= share <= 0.65 else ""))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
