#!/usr/bin/env python3
"""Sanity of the batched arena: a net against itself must split ~50/50 whatever it is; swapping A and B must swap the score."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ataxxzero_amd import arena, model
nets = {s: model.random_init(12, 128, seed=s) for s in (1, 2)}
for a, b in ((1, 1), (2, 2), (1, 2), (2, 1)):
    m = arena.Match(nets[a], nets[b], visits=int(os.environ.get("V", "60")), games=256)
    games = []
    while len(games) < 400:
        m.run(200)
        games += m.drain()
    wa = sum(1 for g in games if (g["result"] == 1) == (g["white"] == "a") and g["result"] in (1, 2))
    wb = sum(1 for g in games if (g["result"] == 1) != (g["white"] == "a") and g["result"] in (1, 2))
    xw = sum(1 for g in games if g["result"] == 1)
    print("net %d vs net %d: A %d - B %d of %d   (x wins %d, mean plies %.0f)" % (a, b, wa, wb, len(games), xw,
          sum(len(g["moves"]) for g in games) / len(games)))
    m.close()
