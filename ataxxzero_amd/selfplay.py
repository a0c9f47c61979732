"""Host side of the batched self-play generator: configuration, the device-resident
generation loop and the reference-format game writers.

Mirrors what accelerated_generate_games.py:24-83 does around the C++ workers: load the
net, keep thousands of games in flight, append finished games as JSON lines (flushing
per batch so looper.py's line counter, looper.py:5-12, sees them), print a `Rate:` line.
"""
import json
import os
import re
import time

import numpy as np

from . import link, model

START_FEN_SELFPLAY = "x5o/7/3-3/2-1-2/3-3/7/o5x x"   # cpp/self_play_client.cpp:23
START_FEN_PLAIN = "x5o/7/7/7/7/7/o5x x"               # ataxx_rules.py:44-50, cpp/ataxx.cpp:11 family
DEFAULT_SEED = 20260101


def parse_fen(fen):
    """-> (x, o, blockers, turn) bitboards; FEN rows run from rank 7 down (cpp/ataxx.cpp:14-90)."""
    parts = fen.split()
    x = o = bl = 0
    sq = 42
    for c in parts[0]:
        if c in "xX":
            x |= 1 << sq
            sq += 1
        elif c in "oO":
            o |= 1 << sq
            sq += 1
        elif c == "-":
            bl |= 1 << sq
            sq += 1
        elif c in "1234567":
            sq += int(c)
        elif c == "/":
            sq -= 14
        else:
            raise ValueError("bad FEN %r" % fen)
    turn = 1 if len(parts) > 1 and parts[1] in "oO" else 0
    return x, o, bl, turn


def make_config(games, visits, seed=DEFAULT_SEED, fen=START_FEN_SELFPLAY, max_plies=400, edges_per_node=96,
                c_puct=1.0, dirichlet_alpha=0.15, dirichlet_weight=0.25):
    """Search constants default to cpp/self_play_client.cpp:31-34."""
    x, o, bl, turn = parse_fen(fen)
    return link.Config(games=games, visits=visits, max_plies=max_plies, edges_per_node=edges_per_node,
                       c_puct=c_puct, dirichlet_alpha=dirichlet_alpha, dirichlet_weight=dirichlet_weight,
                       start_turn=turn, seed=seed, start_x=x, start_o=o, blockers=bl)


def process_index_from_path(path):
    """looper.py:70-74 names the per-process files model-%03i-%i.json; the trailing index
    selects the GPU and the RNG stream so N generator processes shard over N GPUs."""
    m = re.search(r"-(\d+)\.json$", os.path.basename(path))
    return int(m.group(1)) if m else 0


def select_device(index):
    n = link.require_gpu()
    dev = int(os.environ["AZH_DEVICE"]) if "AZH_DEVICE" in os.environ else index % n
    link.check(link.load().azh_set_device(dev))
    return dev


class SelfPlay:
    """`games` concurrent MCTS self-play games on one GPU with the built-in net."""

    def __init__(self, conv_weights, bn_params, games, visits, dtype="bf16", seed=DEFAULT_SEED,
                 fen=START_FEN_SELFPLAY, **cfg):
        self.dtype = link.DTYPES[dtype]
        self.net = link.Net(conv_weights, bn_params, model.BN_EPSILON)
        self.engine = link.Engine(make_config(games, visits, seed=seed, fen=fen, **cfg))
        self.games = games

    def run(self, iterations):
        self.engine.run(self.net, iterations, self.dtype)

    def drain(self):
        return self.engine.drain_json()

    def stats(self):
        return self.engine.stats()

    def close(self):
        self.engine.close()
        self.net.close()


def python_move(mv):
    """u16 move -> the Python generator's move value (generate_games.py:51 stores the
    ataxx_rules tuple; json turns it into nested lists): ["c",[x,y]] / [[x0,y0],[x1,y1]]."""
    frm, to = mv & 0xFF, mv >> 8
    end = [to % 7, 6 - to // 7]
    if frm == to:
        return ["c", end]
    return [[frm % 7, 6 - frm // 7], end]


def board_cells(x, o):
    """49 ints, index x + 7*y with y = 0 at rank 7 (ataxx_rules.py:74-80)."""
    return [1 if (x >> (c + 7 * (6 - r))) & 1 else (2 if (o >> (c + 7 * (6 - r))) & 1 else 0)
            for r in range(7) for c in range(7)]


def random_play_entries(n_games, seed, max_plies=400):
    """Uniform-random games on the GPU in generate_games.py's entry shape (:20,:50-51,:68):
    {"boards": [...], "moves": [...], "result": r}; unfinished games have result None."""
    x, o, bl, turn = parse_fen(START_FEN_PLAIN)
    plies, results, boards, moves = link.random_play(n_games, seed, x, o, bl, turn, max_plies)
    out = []
    for g in range(n_games):
        n = int(plies[g])
        out.append({
            "boards": [board_cells(int(boards[g, p, 0]), int(boards[g, p, 1])) for p in range(n)],
            "moves": [python_move(int(moves[g, p])) for p in range(n)],
            "result": int(results[g]) if results[g] else None,
        })
    return out
