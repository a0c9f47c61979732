#!/usr/bin/env python3
"""What the reduced-precision tower does to the SEARCH (not just to the logits): the same mid-game positions are
searched with the f32, bf16 and f16 towers — same Philox seeds, same sims/move, no Dirichlet noise so the evaluator
is the only difference — and the root visit distributions are compared.

    python tools/precision_in_the_loop.py [--positions 256] [--visits 400] [--blocks 12] [--network trained.npy]

Prints one JSON object: per dtype, top-1 (most visited move) agreement with the f32 search, mean / p95 / max
KL(f32 || dtype) of the root visit distributions and mean total-variation distance, plus max |dlogit| / |dvalue| on
the root positions.  Positions come from GPU random play (no-blocker start family with the self-play blockers).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ataxxzero_amd import link, model, selfplay  # noqa: E402


def midgame_positions(n, seed, blockers):
    x, o, _, turn = selfplay.parse_fen(selfplay.START_FEN_SELFPLAY)
    plies, results, boards, moves = link.random_play(max(64, n), seed, x, o, blockers, turn, 300)
    rng = np.random.default_rng(seed)
    out = []
    g = 0
    while len(out) < n:
        lo, hi = 8, max(int(plies[g % len(plies)]) - 6, 9)
        ply = int(rng.integers(lo, hi))
        out.append((int(boards[g % len(plies), ply, 0]), int(boards[g % len(plies), ply, 1]), ply % 2))
        g += 1
    return out


def root_distribution(net, dtype, pos, blockers, visits, seed):
    """Search every position once: one single-slot engine per position, all enqueued before any is waited for (every
    engine has its own stream, so the searches overlap on the GPU); stop when the move is due."""
    engines = []
    for i, (x, o, turn) in enumerate(pos):
        cfg = link.Config(games=1, visits=visits, max_plies=400, edges_per_node=96, c_puct=1.0, dirichlet_alpha=0.15,
                          dirichlet_weight=0.0, start_turn=turn, seed=seed + i, start_x=x, start_o=o, blockers=blockers,
                          flags=0, select_budget=0)
        e = link.Engine(cfg)
        e.run(net, visits + 1, dtype)     # one root evaluation + `visits` steps
        engines.append(e)
    out = []
    for e in engines:
        e.sync()
        s = e.game_state(0)
        assert s.ply == 0 and s.phase == 2 and s.root_visits == visits, s.as_tuple()
        boards, info, edges, moves = e.tree(0)
        first, m = int(info[0, 0]), int(info[0, 1] & 0xFFFF)
        out.append((moves[first:first + m].copy(), edges[first:first + m, 1].astype(np.float64)))
        e.close()
    return out


def compare(ref, other):
    top1, kls, tvs = [], [], []
    for (mv_a, n_a), (mv_b, n_b) in zip(ref, other):
        assert (mv_a == mv_b).all()
        p, q = n_a / n_a.sum(), n_b / n_b.sum()
        top1.append(int(np.argmax(n_a) == np.argmax(n_b)))
        eps = 0.5 / n_a.sum()             # half a visit: a move one search never tried does not make KL infinite
        kls.append(float(np.sum(p * np.log((p + eps) / (q + eps)))))
        tvs.append(float(0.5 * np.abs(p - q).sum()))
    return {"top1_agreement": float(np.mean(top1)), "kl_mean": float(np.mean(kls)), "kl_median": float(np.median(kls)), "kl_p95": float(np.percentile(kls, 95)),
            "kl_max": float(np.max(kls)), "tv_mean": float(np.mean(tvs))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=256)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--network", help="weights to search with (.npy); default: a random-init net, seed 1")
    args = ap.parse_args()
    link.require_gpu()
    blockers = selfplay.parse_fen(selfplay.START_FEN_SELFPLAY)[2]
    conv, bn = model.load_model(args.network) if args.network else model.random_init(args.blocks, 128, seed=1)
    net = link.Net(conv, bn)
    pos = midgame_positions(args.positions, args.seed, blockers)
    lb = np.array([[x, o] if t == 0 else [o, x] for x, o, t in pos], dtype=np.uint64)
    p32, v32 = net.forward(lb, blockers, link.DTYPE_F32)
    report = {"positions": len(pos), "visits": args.visits,
              "net": os.path.basename(args.network) if args.network else "%dx128 random-init seed 1" % args.blocks,
              "logit_scale": float(np.abs(p32).max())}
    ref = root_distribution(net, link.DTYPE_F32, pos, blockers, args.visits, args.seed)
    # the search's own noise floor: f32 against f32 with nothing changed must be exact
    again = root_distribution(net, link.DTYPE_F32, pos[:16], blockers, args.visits, args.seed)
    report["f32_repeat_identical"] = all((a[1] == b[1]).all() for a, b in zip(ref[:16], again))
    for name, dt in (("bf16", link.DTYPE_BF16), ("f16", link.DTYPE_F16)):
        p, v = net.forward(lb, blockers, dt)
        r = compare(ref, root_distribution(net, dt, pos, blockers, args.visits, args.seed))
        r.update(max_abs_dlogit=float(np.abs(p - p32).max()), max_abs_dvalue=float(np.abs(v - v32).max()),
                 mean_abs_dlogit=float(np.abs(p - p32).mean()))
        report[name] = r
    print(json.dumps(report))


if __name__ == "__main__":
    main()
