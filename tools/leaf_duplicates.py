#!/usr/bin/env python3
"""How many leaf evaluations repeat a position: within a batch (other games), and within a game's own search (transpositions
and re-visited positions) — the share an evaluation cache like engine.py's NNEvaluator.cache (engine.py:127-234) could save.
Steady-state batch (profiles/round2_steady_state_positions.npz), 12x128 bf16 net, 400 sims/move, step-wise API.
Bounds for the per-game share: `since_move` forgets a game's positions at every move (under-counts: the kept subtree
is forgotten too), `since_start` never forgets (over-counts: discarded branches stay)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ataxxzero_amd import link, model, selfplay  # noqa: E402

G = int(os.environ.get("GAMES", "1024"))
ITERS = int(os.environ.get("ITERS", "1500"))
conv, bn = model.random_init(12, 128, seed=1)
sp = selfplay.SelfPlay(conv, bn, games=G, visits=400, dtype="bf16", seed=selfplay.DEFAULT_SEED, select_budget=48)
snap = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                            "round2_steady_state_positions.npz"))
pick = np.random.default_rng(1).integers(0, len(snap["plies"]), size=G)
sp.set_positions(snap["boards"][pick], snap["plies"][pick])
sp.run(600)
sp.sync()
e = sp.engine
since_move = [set() for _ in range(G)]
since_start = [set() for _ in range(G)]
ply = [e.game_state(g).ply for g in range(G)]
tot = batch_dup = hit_move = hit_start = roots = 0
for it in range(ITERS):
    e.select()
    need, lb = e.leaves()
    idx = np.nonzero(need)[0]
    keys = [(int(lb[g, 0]), int(lb[g, 1])) for g in idx]
    batch_dup += len(keys) - len(set(keys))
    if it % 50 == 0:   # plies change rarely; refresh the per-game ply now and then
        for g in range(G):
            p = e.game_state(g).ply
            if p != ply[g]:
                ply[g] = p
                since_move[g].clear()
    for g, k in zip(idx, keys):
        tot += 1
        hit_move += k in since_move[g]
        hit_start += k in since_start[g]
        since_move[g].add(k)
        since_start[g].add(k)
    e.eval(sp.net, sp.dtype)
    e.backup()
print("games %d, iterations %d: %d leaf evaluations; duplicates inside a batch %.2f%%; repeats of a position the same game "
      "evaluated before: %.1f%% (forgetting at every move) .. %.1f%% (never forgetting)" % (
          G, ITERS, tot, 100.0 * batch_dup / tot, 100.0 * hit_move / tot, 100.0 * hit_start / tot))
