#!/usr/bin/env python3
"""Per-level cost of the PUCT descent: host-timed k_select launches against the deepest path of the launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ataxxzero_amd import link, model, selfplay
conv, bn = model.random_init(12, 128, seed=1)
for G in [int(a) for a in sys.argv[1:]] or [4096, 64]:
    sp = selfplay.SelfPlay(conv, bn, games=G, visits=400, dtype="bf16")
    sp.set_visits(16); sp.run(2500); sp.set_visits(400); sp.run(700); sp.sync()
    e = sp.engine
    xs, ys, tot = [], [], []
    for it in range(150):
        t0 = time.perf_counter()
        e.select()
        dt = time.perf_counter() - t0
        pls = np.array([e.game_state(g).path_len for g in range(G)])
        pl = pls.max()
        tot.append(pls.sum())
        e.eval(sp.net, sp.dtype)
        e.backup()
        xs.append(pl); ys.append(dt * 1e6)
    xs, ys = np.array(xs, float), np.array(ys)
    a, b = np.polyfit(xs, ys, 1)
    print("G=%d: select time ~ %.1f us + %.3f us x deepest path (paths %d..%d, times %.0f..%.0f us)" % (G, b, a, xs.min(), xs.max(), ys.min(), ys.max()))
    A = np.stack([xs, np.array(tot, float), np.ones(len(xs))], axis=1)
    c = np.linalg.lstsq(A, ys, rcond=None)[0]
    print("      2-term fit: %.3f us x deepest + %.5f us x total levels + %.1f us" % (c[0], c[1], c[2]))
    sp.close()
