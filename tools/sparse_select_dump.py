#!/usr/bin/env python3
"""Trees of bench.py's steady state, dumped for tools/microbench/sparse_select.hip (the select-only prototype of sparse
child records): the engine is put where bench.py measures it (positions of
profiles/round2_steady_state_positions.npz, trees grown for --fill iterations at full sims/move), then every game's node
table and its edge records AS THEY LIE IN HBM (azh_engine_tree_raw: prior with the descent's mark, score, visits | child,
the child's range) are written to one binary file:

    int32 games; per game: int32 n_nodes, n_edges, root_visits; uint32 info[n_nodes][4]; uint32 edges[n_edges][4]

    python tools/sparse_select_dump.py --out /tmp/azh_trees.bin [--games 4096] [--visits 400]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ataxxzero_amd import link, model, selfplay  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--fill", type=int, default=1000)
    ap.add_argument("--select-budget", type=int, default=48)
    args = ap.parse_args()
    link.require_gpu()
    conv, bn = model.random_init(12, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=args.games, visits=args.visits, dtype="bf16", select_budget=args.select_budget, streams=1)
    snap = np.load(os.path.join(ROOT, "profiles", "round2_steady_state_positions.npz"))
    rng = np.random.default_rng(1)
    pick = rng.permutation(len(snap["plies"])) if args.games == len(snap["plies"]) else rng.integers(0, len(snap["plies"]), args.games)
    sp.set_positions(snap["boards"][pick], snap["plies"][pick])
    done = 0
    while done < args.fill:
        sp.run(250)
        sp.drain()
        done += 250
    sp.sync()
    e = sp.engines[0]
    nodes = edges = 0
    with open(args.out, "wb") as f:
        f.write(np.array([args.games], dtype=np.int32).tobytes())
        for g in range(args.games):
            s = e.game_state(g)
            _, info, _, _ = e.tree(g)
            raw = e.tree_raw(g)
            f.write(np.array([s.n_nodes, s.n_edges, s.root_visits], dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(info, dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(raw, dtype=np.uint32).tobytes())
            nodes += s.n_nodes
            edges += s.n_edges
    print("%d games, %d nodes, %d edges (%.1f per node), %.2f GB -> %s" % (args.games, nodes, edges, edges / float(nodes),
                                                                         os.path.getsize(args.out) / 1e9, args.out))
    sp.close()


if __name__ == "__main__":
    main()
