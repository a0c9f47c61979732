#!/usr/bin/env python3
"""Distribution of PUCT descent depths over the games of a warm self-play batch (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ataxxzero_amd import link, model, selfplay
conv, bn = model.random_init(12, 128, seed=1)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sp = selfplay.SelfPlay(conv, bn, games=G, visits=400, dtype="bf16")
sp.set_visits(16); sp.run(2500); sp.set_visits(400); sp.run(700); sp.sync()
e = sp.engine
for rep in range(3):
    sp.run(50); sp.sync()
    e.select()
    pl = np.array([e.game_state(g).path_len for g in range(G)])
    nn = np.array([e.game_state(g).n_nodes for g in range(G)])
    print("path_len: mean %.1f  p50 %d  p90 %d  p99 %d  max %d   nodes mean %.0f" % (
        pl.mean(), np.percentile(pl, 50), np.percentile(pl, 90), np.percentile(pl, 99), pl.max(), nn.mean()))
    need, lb = e.leaves()
    lg = np.zeros((G, 833), np.float32); v = np.zeros(G, np.float32)
    e.set_evals(lg, v); e.backup()
