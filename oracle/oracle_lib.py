"""ORACLE — TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/_build/liboracle.so.

Built on demand with gcc from oracle/*.c (see oracle/Makefile).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")

STAT_NAMES = ["steps", "nn_evals", "levels", "children", "new_moves", "plies", "games", "dropped",
              "edge_overflow", "reroot_nodes", "reroot_edges", "cache_hits"]
STAT_COUNT = 16
LEAF_NONE, LEAF_EVAL, LEAF_TERMINAL, LEAF_ROOT, LEAF_DESCENT = 0, 1, 2, 3, 4
FLAG_NO_REUSE, FLAG_TIE_FIRST, FLAG_PY_POSTERIOR, FLAG_SAMPLE_POW5, FLAG_KEEP_UNFINISHED, FLAG_TWO_NETS = 1, 2, 4, 8, 16, 32
FLAG_ARENA = 63
FLAG_ONE_RANDOM_MOVE = 64
FLAG_EVAL_CACHE = 256


class Pos(ctypes.Structure):
    _fields_ = [("pieces", ctypes.c_uint64 * 2), ("blockers", ctypes.c_uint64),
                ("turn", ctypes.c_int32), ("ply", ctypes.c_int32)]


class Config(ctypes.Structure):
    _fields_ = [("games", ctypes.c_int32), ("visits", ctypes.c_int32), ("max_plies", ctypes.c_int32),
                ("edges_per_node", ctypes.c_int32), ("c_puct", ctypes.c_float),
                ("dirichlet_alpha", ctypes.c_float), ("dirichlet_weight", ctypes.c_float),
                ("start_turn", ctypes.c_int32), ("seed", ctypes.c_uint64), ("start_x", ctypes.c_uint64),
                ("start_o", ctypes.c_uint64), ("blockers", ctypes.c_uint64),
                ("flags", ctypes.c_uint32), ("select_budget", ctypes.c_uint32)]


class GameState(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("phase", "arena", "n_nodes", "n_edges", "ply", "root_visits",
                                              "leaf_kind", "leaf_node", "path_len")] + [("uid", ctypes.c_uint32)]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def build(force=False):
    if os.environ.get("ORACLE_LIB"):
        return os.environ["ORACLE_LIB"]  # e.g. the sanitizer build (`make -C oracle asan`, see the Makefile)
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = ctypes.CDLL(build())
    u64, i32, u32, f32 = ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_float
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    P = ctypes.POINTER
    sig = {
        "orc_singles": (u64, [ctypes.c_int]), "orc_doubles": (u64, [ctypes.c_int]),
        "orc_single_jump_bb": (u64, [u64]),
        "orc_set_fen": (ctypes.c_int, [P(Pos), cp]), "orc_fen": (ctypes.c_int, [P(Pos), cp, ctypes.c_int]),
        "orc_movegen": (ctypes.c_int, [P(Pos), vp]), "orc_makemove": (None, [P(Pos), ctypes.c_int, ctypes.c_int]),
        "orc_pass": (None, [P(Pos)]), "orc_result": (ctypes.c_int, [P(Pos), vp, P(ctypes.c_int)]),
        "orc_perft": (u64, [P(Pos), ctypes.c_int]), "orc_move_string": (ctypes.c_int, [ctypes.c_uint16, cp]),
        "orc_policy_index": (ctypes.c_int, [ctypes.c_uint16]), "orc_features": (None, [P(Pos), vp]),
        "orc_board_cells": (None, [P(Pos), vp]),
        "orc_engine_create": (vp, [P(Config)]), "orc_engine_destroy": (None, [vp]),
        "orc_engine_node_cap": (ctypes.c_int, [vp]), "orc_engine_edge_cap": (ctypes.c_int, [vp]),
        "orc_engine_set_visits": (None, [vp, ctypes.c_int]),
        "orc_engine_set_positions": (None, [vp, vp, vp]),
        "orc_engine_set_game_limit": (None, [vp, ctypes.c_int64]),
        "orc_engine_select": (ctypes.c_int, [vp, vp]), "orc_engine_leaf_boards": (None, [vp, vp]),
        "orc_engine_leaf_features": (None, [vp, ctypes.c_int, vp]),
        "orc_engine_backup": (None, [vp, vp, vp]),
        "orc_engine_game_state": (None, [vp, ctypes.c_int, P(GameState)]),
        "orc_engine_tree": (None, [vp, ctypes.c_int, vp, vp, vp, vp]),
        "orc_engine_stats": (None, [vp, vp]),
        "orc_engine_pop_game": (ctypes.c_int64, [vp, vp, ctypes.c_int64]),
        "orc_engine_pending_games": (ctypes.c_int, [vp]),
        "orc_probe_expf": (f32, [f32]), "orc_probe_logf": (f32, [f32]),
        "orc_probe_gamma": (f32, [f32, u64, u32, u32, u32]),
        "orc_probe_philox": (None, [u64, u32, u32, u32, u32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ---------------------------------------------------------------- rules helpers

def pos_from_fen(fen):
    p = Pos()
    rc = lib().orc_set_fen(ctypes.byref(p), fen.encode())
    if rc != 0:
        raise ValueError("bad fen %r (%d)" % (fen, rc))
    return p


def fen(p):
    buf = ctypes.create_string_buffer(80)
    lib().orc_fen(ctypes.byref(p), buf, 80)
    return buf.value.decode()


def movegen(p):
    moves = np.zeros(256, dtype=np.uint16)
    n = lib().orc_movegen(ctypes.byref(p), moves.ctypes.data)
    return moves[:n].copy()


def move_string(m):
    buf = ctypes.create_string_buffer(8)
    lib().orc_move_string(int(m), buf)
    return buf.value.decode()


def move_from_string(s):
    sq = lambda t: (ord(t[0]) - 97) + 7 * (int(t[1]) - 1)
    if len(s) == 2:
        return sq(s) | (sq(s) << 8)
    return sq(s[:2]) | (sq(s[2:]) << 8)


def result(p):
    return lib().orc_result(ctypes.byref(p), None, None)


def perft(p, depth):
    return int(lib().orc_perft(ctypes.byref(p), depth))


def features(p):
    out = np.zeros((7, 7, 4), dtype=np.float32)
    lib().orc_features(ctypes.byref(p), out.ctypes.data)
    return out


def board_cells(p):
    out = np.zeros(49, dtype=np.int32)
    lib().orc_board_cells(ctypes.byref(p), out.ctypes.data)
    return out


# ---------------------------------------------------------------- search engine

START_FEN_SELFPLAY = "x5o/7/3-3/2-1-2/3-3/7/o5x x"   # cpp/self_play_client.cpp:23
START_FEN_PLAIN = "x5o/7/7/7/7/7/o5x x"               # ataxx_rules.py:44-50


def make_config(games, visits, seed=20260101, fen_str=START_FEN_SELFPLAY, max_plies=400,
                edges_per_node=96, c_puct=1.0, alpha=0.15, weight=0.25, flags=0, select_budget=0):
    p = pos_from_fen(fen_str)
    return Config(games=games, visits=visits, max_plies=max_plies, edges_per_node=edges_per_node,
                  c_puct=c_puct, dirichlet_alpha=alpha, dirichlet_weight=weight, start_turn=p.turn,
                  seed=seed, start_x=p.pieces[0], start_o=p.pieces[1], blockers=p.blockers, flags=flags,
                  select_budget=select_budget)


def parse_game_record(buf):
    """Decode one packed finished-game record (see mcts_oracle.h) into the
    reference's JSON entry shape (cpp/self_play_client.cpp:512,565-578)."""
    hdr = np.frombuffer(buf[:16], dtype=np.int32)
    slot, uid, plies, res = (int(v) for v in hdr)
    off = 16
    boards, moves, dists = [], [], []
    for _ in range(plies):
        x, o = (int(v) for v in np.frombuffer(buf[off:off + 16], dtype=np.uint64))
        mv, nd = (int(v) for v in np.frombuffer(buf[off + 16:off + 20], dtype=np.uint16))
        off += 24
        ents = np.frombuffer(buf[off:off + 4 * nd], dtype=np.uint32)
        off += 4 * nd
        p = Pos()
        p.pieces[0], p.pieces[1] = x, o
        boards.append([int(v) for v in board_cells(p)])
        moves.append(move_string(mv))
        total = int(sum(int(e) >> 16 for e in ents))
        dists.append({move_string(int(e) & 0xFFFF): (int(e) >> 16) / total for e in ents})
    entry = {"boards": boards, "dists": dists, "moves": moves, "result": res & 0xFF}
    if (res >> 8) & 0x3FFFFF:
        entry["random_ply"] = ((res >> 8) & 0x3FFFFF) - 1
    # partial: the game began at a loaded position (set_positions); its record starts there
    return {"slot": slot, "uid": uid & 0xFFFFFFFF, "entry": entry, "partial": bool((res >> 30) & 1)}


class Engine:
    def __init__(self, cfg):
        self.cfg = cfg
        self.G = cfg.games
        self.h = lib().orc_engine_create(ctypes.byref(cfg))
        self.node_cap = lib().orc_engine_node_cap(self.h)
        self.edge_cap = lib().orc_engine_edge_cap(self.h)

    def close(self):
        if self.h:
            lib().orc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_visits(self, visits):
        lib().orc_engine_set_visits(self.h, visits)

    def set_game_limit(self, games):
        lib().orc_engine_set_game_limit(self.h, int(games))

    def set_positions(self, boards, plies):
        boards = np.ascontiguousarray(boards, dtype=np.uint64).reshape(self.G, 2)
        plies = np.ascontiguousarray(plies, dtype=np.int32).reshape(self.G)
        lib().orc_engine_set_positions(self.h, boards.ctypes.data, plies.ctypes.data)

    def select(self):
        need = np.zeros(self.G, dtype=np.int32)
        n = lib().orc_engine_select(self.h, need.ctypes.data)
        return n, need

    def leaf_boards(self):
        out = np.zeros((self.G, 2), dtype=np.uint64)
        lib().orc_engine_leaf_boards(self.h, out.ctypes.data)
        return out

    def leaf_features(self, g):
        out = np.zeros((7, 7, 4), dtype=np.float32)
        lib().orc_engine_leaf_features(self.h, g, out.ctypes.data)
        return out

    def backup(self, logits, values):
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        assert logits.shape == (self.G, 833) and values.shape == (self.G,)
        lib().orc_engine_backup(self.h, logits.ctypes.data, values.ctypes.data)

    def game_state(self, g):
        s = GameState()
        lib().orc_engine_game_state(self.h, g, ctypes.byref(s))
        return s

    def tree(self, g):
        s = self.game_state(g)
        boards = np.zeros((s.n_nodes, 2), dtype=np.uint64)
        info = np.zeros((s.n_nodes, 4), dtype=np.uint32)
        edges = np.zeros((s.n_edges, 4), dtype=np.uint32)
        moves = np.zeros(s.n_edges, dtype=np.uint16)
        lib().orc_engine_tree(self.h, g, boards.ctypes.data, info.ctypes.data, edges.ctypes.data,
                              moves.ctypes.data)
        return boards, info, edges, moves

    def stats(self):
        out = np.zeros(STAT_COUNT, dtype=np.uint64)
        lib().orc_engine_stats(self.h, out.ctypes.data)
        return {n: int(out[i]) for i, n in enumerate(STAT_NAMES)}

    def pop_games(self, partial=False):
        """Finished games since the last call.  Games that began at a loaded position are left out unless `partial` (the
        self-play callers of the HIP engine do not write them; the arena hands them out)."""
        games = []
        buf = np.zeros(1 << 20, dtype=np.uint8)
        while lib().orc_engine_pending_games(self.h):
            n = lib().orc_engine_pop_game(self.h, buf.ctypes.data, buf.nbytes)
            if n == 0:
                buf = np.zeros(buf.nbytes * 2, dtype=np.uint8)
                continue
            rec = parse_game_record(bytes(buf[:n]))
            if partial or not rec["partial"]:
                games.append(rec)
        return games
