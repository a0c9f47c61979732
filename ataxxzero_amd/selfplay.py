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
                c_puct=1.0, dirichlet_alpha=0.15, dirichlet_weight=0.25, flags=0, select_budget=0):
    """Search constants default to cpp/self_play_client.cpp:31-34."""
    x, o, bl, turn = parse_fen(fen)
    return link.Config(games=games, visits=visits, max_plies=max_plies, edges_per_node=edges_per_node,
                       c_puct=c_puct, dirichlet_alpha=dirichlet_alpha, dirichlet_weight=dirichlet_weight,
                       start_turn=turn, seed=seed, start_x=x, start_o=o, blockers=bl, flags=flags,
                       select_budget=select_budget)


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
    """`games` concurrent MCTS self-play games on one GPU with the built-in net.

    With `streams` = 2 the games are split into two half-batches, each with its own engine
    and HIP stream, sharing one set of packed weights — the reference's double buffer
    (cpp/self_play_client.cpp:593-600, two fill buffers of `buffer_entries` rows): while one
    half runs its tree kernels or the ramp or tail of its tower launch, the other half's tower fills
    the idle CUs (+10 % at 4096 games, +24 % at 2048, nothing from 16384 on:
    profiles/round4_half_batches_and_reserved_cus.txt; bench.py's and the generator's default).
    The halves are independent game shards (distinct Philox streams, their own uids)."""

    def __init__(self, conv_weights, bn_params, games, visits, dtype="bf16", seed=DEFAULT_SEED,
                 fen=START_FEN_SELFPLAY, streams=1, **cfg):
        self.dtype = link.DTYPES[dtype]
        self.net = link.Net(conv_weights, bn_params, model.BN_EPSILON)
        if streams < 1 or games < streams:
            raise ValueError("games (%d) must be at least the number of half-batches (%d)" % (games, streams))
        # (an uneven split gives the first games % streams engines one game more)
        sizes = [games // streams + (1 if i < games % streams else 0) for i in range(streams)]
        self.engines = [link.Engine(make_config(sizes[i], visits, seed=seed + 1000003 * i, fen=fen, **cfg))
                        for i in range(streams)]
        self.engine = self.engines[0]
        self.games = games

    def run(self, iterations):
        # every engine's whole run is enqueued on its own stream (the calls are asynchronous): the half-batches then
        # advance independently on the GPU, one's tree phase and tower tail filled by the other's tower.  The engines'
        # iterations are enqueued in turn (azh_engines_run), so that the second half starts with the first launches
        # enqueued and not a run's worth of host time later (AZH_ENQUEUE_SEQUENTIAL=1: engine after engine, rounds 4-5).
        if len(self.engines) > 1 and os.environ.get("AZH_ENQUEUE_SEQUENTIAL", "0") in ("", "0"):
            link.run_engines(self.engines, self.net, iterations, self.dtype)
            return
        for e in self.engines:
            e.run(self.net, iterations, self.dtype)

    def sync(self):
        for e in self.engines:
            e.sync()

    def set_visits(self, visits):
        for e in self.engines:
            e.set_visits(visits)

    def set_thin_batches(self, mode):
        """1: the towers run one board per workgroup (a handful of leaves per iteration: the tail of a run under a game
        limit); 0: the 3-board workgroups; -1: by each engine's size (link.Engine.set_thin_batches)."""
        for e in self.engines:
            e.set_thin_batches(mode)

    def set_positions(self, boards, plies):
        lo = 0
        for e in self.engines:
            e.set_positions(boards[lo:lo + e.G], plies[lo:lo + e.G])
            lo += e.G

    def positions(self):
        """(root boards packed (G,2) u64, plies (G,)) of all slots: what set_positions takes."""
        boards, plies = [], []
        for e in self.engines:
            for g in range(e.G):
                boards.append(e.tree(g)[0][0])
                plies.append(e.game_state(g).ply)
        return np.asarray(boards, dtype=np.uint64), np.asarray(plies, dtype=np.int32)

    def set_emit_order(self, by_uid):
        for e in self.engines:
            e.set_emit_order(by_uid)

    def set_game_limit(self, games):
        """`games` games in all, then the slots go idle.  uids are per engine: half-batch i of K plays its own uids
        0 .. ceil((games - i) / K) - 1, so the limits add up to `games` (and may be raised later, like the engine's)."""
        k = len(self.engines)
        if games < k:
            raise ValueError("a game limit of %d needs at most %d half-batches" % (games, games))
        for i, e in enumerate(self.engines):
            e.set_game_limit((games + k - 1 - i) // k)

    def fetch(self):
        """sync + finished games off the device; `drain` then formats them on the host while the GPU does something else"""
        for e in self.engines:
            e.fetch()

    def drain(self):
        lines = []
        for e in self.engines:
            lines += e.drain_json()
        return lines

    def stats(self):
        total = {}
        for e in self.engines:
            for k, v in e.stats().items():
                total[k] = total.get(k, 0) + v
        return total

    def timing_reset(self, enable=True):
        for e in self.engines:
            e.timing_reset(enable)

    def timing(self):
        total = {}
        for e in self.engines:
            for k, v in e.timing().items():
                total[k] = total.get(k, 0) + v
        return total

    def close(self):
        for e in self.engines:
            e.close()
        self.net.close()


def python_move(mv):
    """u16 move -> the Python generator's move value (generate_games.py:51 stores the
    ataxx_rules tuple; json turns it into nested lists): ["c",[x,y]] / [[x0,y0],[x1,y1]]."""
    if mv == 0xFFFF:
        return "pass"
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
