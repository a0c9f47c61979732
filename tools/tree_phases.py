#!/usr/bin/env python3
"""Warm self-play batch driven through the step-wise API (k_select / tower / k_backup / k_advance as separate
launches), for a per-phase kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/tree_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ataxxzero_amd import link, model, selfplay
conv, bn = model.random_init(12, 128, seed=1)
G = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("GAMES", "4096"))
sp = selfplay.SelfPlay(conv, bn, games=G, visits=400, dtype="bf16")
sp.set_visits(16); sp.run(2500); sp.set_visits(400); sp.run(700); sp.sync()
e = sp.engine
for it in range(300):
    e.select()
    e.eval(sp.net, sp.dtype)
    e.backup()
print("done", e.stats()["plies"])
