#!/usr/bin/env python3
"""Would the config-5 match gain from two half-batches in flight?  1000 games as ONE engine against two engines of 500 games
(two arena.Match objects on their own streams, rounds enqueued alternately): wall time until 488 / 90 % / 99 % of the games are
over.  A probe, not the product: the two matches here are independent cohorts."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ataxxzero_amd import arena, model, selfplay  # noqa: E402


def play(nets, sizes, visits=100, dtype="f16", thin_after=None):
    ms = [arena.Match(nets[0], nets[1], visits, games=g, dtype=dtype, seed=selfplay.DEFAULT_SEED + i) for i, g in enumerate(sizes)]
    total = sum(sizes)
    for m, g in zip(ms, sizes):
        m.set_game_limit(g)
        m.run(5)
    for m in ms:
        m.engine.sync()
    t0 = time.perf_counter()
    done, marks, rounds = 0, {}, 0
    for m in ms:
        m.run(50)
    while done < total:
        for m in ms:
            m.fetch()
            m.run(50)
            done += len(m.drain())
        rounds += 1
        for name, frac in (("488", 0.488), ("90%", 0.90), ("99%", 0.99)):
            if name not in marks and done >= frac * total:
                marks[name] = (time.perf_counter() - t0, rounds * 50)
    dt = time.perf_counter() - t0
    for m in ms:
        m.close()
    return dt, marks


def main():
    selfplay.select_device(0)
    for seeds in ((1, 2), (27, 28)):
        nets = [model.random_init(12, 128, seed=s) for s in seeds]
        for sizes in ((1000,), (500, 500), (1000,), (500, 500)):
            dt, marks = play(nets, sizes)
            print("seeds %s, engines %-10s: match %.2f s | %s" % (seeds, sizes, dt, " | ".join(
                "%s of the games after %.2f s (%d iterations)" % (k, v[0], v[1]) for k, v in marks.items())))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
