"""ctypes binding of libataxxzero_hip.so (include/ataxxzero_hip.h).

Mirror of the reference's link.py (link.py:6-32): the same four module-level
callables — launch_threads, get_workload, complete_workload, shutdown — with the
same argument order, bound to the GPU library instead of
./cpp/self_play_client.so, plus the `azh_*` device-resident entry points.

There is no CPU fallback: if the library is missing it is built with hipcc; if
no MI355X is visible every compute call raises AzhError.
"""
import ctypes
import os

import numpy as np

from . import build as _build

POLICY_SIZE = 833
FEATURE_SIZE = 196
MAX_MOVES = 256
STAT_COUNT = 16
THIN_MAX_GAMES = 512        # AZH_THIN_MAX_GAMES: engines of at most this many game slots evaluate with one board per workgroup
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
DTYPES = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32, "bf16": DTYPE_BF16,
          "f16": DTYPE_F16, "fp16": DTYPE_F16}
LEAF_NONE, LEAF_EVAL, LEAF_TERMINAL, LEAF_ROOT = 0, 1, 2, 3
FLAG_NO_REUSE, FLAG_TIE_FIRST, FLAG_PY_POSTERIOR, FLAG_SAMPLE_POW5, FLAG_KEEP_UNFINISHED, FLAG_TWO_NETS = 1, 2, 4, 8, 16, 32
FLAG_ARENA = 63
FLAG_SYMMETRY_AVG = 128    # nn_evals.py:48-62 on every evaluation
FLAG_ONE_RANDOM_MOVE = 64  # cpp/self_play_client.cpp:515-552 (compile-time variant of the reference client)
FLAG_EVAL_CACHE = 256      # engine.py:127-234: positions a game's search has already evaluated are not evaluated again
STAT_NAMES = ["steps", "nn_evals", "levels", "children", "new_moves", "plies", "games", "dropped",
              "edge_overflow", "reroot_nodes", "reroot_edges", "ring_overflow", "cache_hits", "parked", "reroot_spills"]


class AzhError(RuntimeError):
    pass


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


class Timing(ctypes.Structure):
    _fields_ = [("select_ms", ctypes.c_double), ("net_ms", ctypes.c_double), ("backup_ms", ctypes.c_double),
                ("iterations", ctypes.c_int64), ("net_evals", ctypes.c_int64)]


_vp, _i32, _u64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float
_P = ctypes.POINTER

# name -> (restype, argtypes); every symbol include/ataxxzero_hip.h declares
SIGNATURES = {
    "azh_last_error": (ctypes.c_char_p, []),
    "azh_device_count": (ctypes.c_int, []),
    "azh_set_device": (ctypes.c_int, [ctypes.c_int]),
    "azh_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "azh_perft": (ctypes.c_int, [_u64, _u64, _u64, ctypes.c_int, ctypes.c_int, _P(_u64)]),
    "azh_rules_batch": (ctypes.c_int, [ctypes.c_int, _vp, _u64, _vp, _vp, _vp]),
    "azh_makemove_batch": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp]),
    "azh_features_batch": (ctypes.c_int, [ctypes.c_int, _vp, _u64, _vp]),
    "azh_random_play": (ctypes.c_int, [ctypes.c_int, _u64, _u64, _u64, _u64, ctypes.c_int, ctypes.c_int,
                                       _vp, _vp, _vp, _vp]),
    "azh_probe_detmath": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _u64, _vp]),
    "azh_net_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _f32, _P(_vp)]),
    "azh_net_destroy": (None, [_vp]),
    "azh_net_forward": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _u64, _vp, _vp]),
    "azh_net_forward_sym": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _u64, _vp, _vp]),
    "azh_net_bench": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P(_f32)]),
    "azh_net_forward_thin": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _u64, _vp, _vp]),
    "azh_net_bench_thin": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P(_f32)]),
    "azh_net_stamps": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "azh_engine_create": (ctypes.c_int, [_P(Config), _P(_vp)]),
    "azh_engine_destroy": (None, [_vp]),
    "azh_engine_node_cap": (ctypes.c_int, [_vp]),
    "azh_engine_edge_cap": (ctypes.c_int, [_vp]),
    "azh_engine_select": (ctypes.c_int, [_vp, _P(_i32)]),
    "azh_engine_leaves": (ctypes.c_int, [_vp, _vp, _vp]),
    "azh_engine_leaf_features": (ctypes.c_int, [_vp, _vp, _vp]),
    "azh_engine_eval": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "azh_engine_set_evals": (ctypes.c_int, [_vp, _vp, _vp]),
    "azh_engine_backup": (ctypes.c_int, [_vp]),
    "azh_engine_run": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "azh_engines_run": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int]),
    "azh_engine_run_arena": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    "azh_engine_sync": (ctypes.c_int, [_vp]),
    "azh_engine_set_visits": (ctypes.c_int, [_vp, ctypes.c_int]),
    "azh_engine_set_thin_batches": (ctypes.c_int, [_vp, ctypes.c_int]),
    "azh_engine_game_state": (ctypes.c_int, [_vp, ctypes.c_int, _P(GameState)]),
    "azh_engine_tree": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "azh_engine_tree_raw": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "azh_engine_stats": (ctypes.c_int, [_vp, _vp]),
    "azh_engine_tree_stamps": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "azh_engine_timing_reset": (ctypes.c_int, [_vp, ctypes.c_int]),
    "azh_engine_timing": (ctypes.c_int, [_vp, _P(Timing)]),
    "azh_engine_fetch": (ctypes.c_int, [_vp]),
    "azh_engine_query": (ctypes.c_int, [_vp]),
    "azh_engine_implicit_fetches": (ctypes.c_longlong, [_vp]),
    "azh_engine_drain_json": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _P(ctypes.c_int64), _P(_i32)]),
    "azh_format_record_json": (ctypes.c_int, [_vp, ctypes.c_int64, _i32, _vp, ctypes.c_int64, _P(ctypes.c_int64)]),
    "azh_engine_set_emit_order": (ctypes.c_int, [_vp, ctypes.c_int]),
    "azh_engine_set_positions": (ctypes.c_int, [_vp, _vp, _vp]),
    "azh_engine_set_game_limit": (ctypes.c_int, [_vp, ctypes.c_int64]),
    # the reference's ABI, link.py:8-32
    "launch_threads": (None, [ctypes.c_char_p, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    "get_workload": (ctypes.c_int, []),
    "complete_workload": (None, [ctypes.c_int, _vp, _vp]),
    "shutdown": (None, []),
}

_dll = None


def library_path():
    return _build.LIB


def load():
    """Load (building first if needed) the HIP library and type its symbols."""
    global _dll
    if _dll is not None:
        return _dll
    path = os.environ.get("AZH_LIB") or _build.build()  # AZH_LIB: load a specific build (A/B runs)
    dll = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(dll, name)
        fn.restype = res
        fn.argtypes = args
    _dll = dll
    return dll


def check(rc):
    if rc != 0:
        raise AzhError("ataxxzero_hip error %d: %s" % (rc, load().azh_last_error().decode(errors="replace")))


def device_count():
    n = load().azh_device_count()
    return max(n, 0)


def require_gpu():
    n = load().azh_device_count()
    if n <= 0:
        raise AzhError("no MI355X / HIP device visible (azh_device_count = %d: %s); this package has no CPU "
                       "fallback" % (n, load().azh_last_error().decode(errors="replace")))
    return n


def pci_bus_id(device):
    buf = ctypes.create_string_buffer(64)
    check(load().azh_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


# ------------------------------------------------------------------ reference ABI (link.py:8-32)

def launch_threads(output_path, visits, fill_buffer1, fill_buffer2, buffer_entries, thread_count):
    load().launch_threads(output_path, visits, fill_buffer1, fill_buffer2, buffer_entries, thread_count)


def get_workload():
    return load().get_workload()


def complete_workload(workload, posteriors, values):
    load().complete_workload(workload, posteriors, values)


def shutdown():
    load().shutdown()


def format_record_json(rec, with_ids=False):
    """The JSON line (bytes, no newline) of one finished-game record — uint32 words as the device loop leaves them in its
    ring — in the reference's entry format (cpp/self_play_client.cpp:565-578,639-641).  Host code only: no GPU needed."""
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    buf = np.zeros(1 << 16, dtype=np.uint8)
    used = ctypes.c_int64(0)
    rc = load().azh_format_record_json(_ptr(rec), rec.size, int(bool(with_ids)), _ptr(buf), buf.nbytes, ctypes.byref(used))
    if rc == -6:
        buf = np.zeros(used.value, dtype=np.uint8)
        rc = load().azh_format_record_json(_ptr(rec), rec.size, int(bool(with_ids)), _ptr(buf), buf.nbytes,
                                           ctypes.byref(used))
    check(rc)
    return bytes(buf[:used.value])


# ------------------------------------------------------------------ rules

def pack_board(x, o, turn):
    return np.array([int(x) | (int(turn) << 63), int(o)], dtype=np.uint64)


def perft(x, o, blockers, turn, depth):
    out = ctypes.c_uint64(0)
    check(load().azh_perft(int(x), int(o), int(blockers), int(turn), int(depth), ctypes.byref(out)))
    return int(out.value)


def rules_batch(boards, blockers):
    boards = np.ascontiguousarray(boards, dtype=np.uint64).reshape(-1, 2)
    n = len(boards)
    moves = np.zeros((n, MAX_MOVES), dtype=np.uint16)
    counts = np.zeros(n, dtype=np.int32)
    results = np.zeros(n, dtype=np.int32)
    check(load().azh_rules_batch(n, _ptr(boards), int(blockers), _ptr(moves), _ptr(counts), _ptr(results)))
    return moves, counts, results


def makemove_batch(boards, moves):
    boards = np.ascontiguousarray(boards, dtype=np.uint64).reshape(-1, 2)
    moves = np.ascontiguousarray(moves, dtype=np.uint16)
    out = np.zeros_like(boards)
    check(load().azh_makemove_batch(len(boards), _ptr(boards), _ptr(moves), _ptr(out)))
    return out


def features_batch(leaf_boards, blockers):
    leaf_boards = np.ascontiguousarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
    out = np.zeros((len(leaf_boards), 7, 7, 4), dtype=np.float32)
    check(load().azh_features_batch(len(leaf_boards), _ptr(leaf_boards), int(blockers), _ptr(out)))
    return out


def random_play(n_games, seed, x, o, blockers, turn, max_plies=400, trace=True):
    plies = np.zeros(n_games, dtype=np.int32)
    results = np.zeros(n_games, dtype=np.int32)
    boards = np.zeros((n_games, max_plies, 2), dtype=np.uint64) if trace else None
    moves = np.zeros((n_games, max_plies), dtype=np.uint16) if trace else None
    check(load().azh_random_play(n_games, int(seed), int(x), int(o), int(blockers), int(turn), max_plies,
                                 _ptr(plies), _ptr(results), _ptr(boards), _ptr(moves)))
    return plies, results, boards, moves


def probe_detmath(kind, values=None, aux=None, seed=0):
    values = None if values is None else np.ascontiguousarray(values, dtype=np.float32)
    aux = None if aux is None else np.ascontiguousarray(aux, dtype=np.uint32)
    n = len(values) if kind in (0, 1) else len(aux) // (3 if kind == 2 else 4)
    out = np.zeros(4 * n if kind == 3 else n, dtype=np.uint32)
    check(load().azh_probe_detmath(kind, n, _ptr(values), _ptr(aux), int(seed), _ptr(out)))
    return out


# ------------------------------------------------------------------ network

class Net:
    """Device-resident policy/value net (model.Network forward, model.py:38-79)."""

    def __init__(self, conv_weights, bn_params, bn_eps=1e-3):
        blocks = (len(conv_weights) - 5) // 2
        filters = int(conv_weights[0].shape[-1])
        if len(conv_weights) != 2 * blocks + 5 or len(bn_params) != 2 * (2 * blocks + 1):
            raise ValueError("weight lists do not match model.py's layout (2B+5 / 2(2B+1) arrays)")
        conv_flat = np.concatenate([np.asarray(a, dtype=np.float32).ravel() for a in conv_weights])
        bn_flat = np.concatenate([np.asarray(a, dtype=np.float32).ravel() for a in bn_params])
        self.blocks, self.filters = blocks, filters
        h = ctypes.c_void_p()
        check(load().azh_net_create(blocks, filters, _ptr(conv_flat), _ptr(bn_flat), bn_eps, ctypes.byref(h)))
        self.h = h

    def forward(self, leaf_boards, blockers, dtype=DTYPE_BF16, thin=False):
        """thin: one board per workgroup (the kernel for a handful of boards, azh_net_forward_thin)."""
        leaf_boards = np.ascontiguousarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
        n = len(leaf_boards)
        logits = np.zeros((n, 7, 7, 17), dtype=np.float32)
        values = np.zeros((n, 1), dtype=np.float32)
        fn = load().azh_net_forward_thin if thin else load().azh_net_forward
        check(fn(self.h, dtype, n, _ptr(leaf_boards), int(blockers), _ptr(logits), _ptr(values)))
        return logits, values

    def forward_sym(self, leaf_boards, blockers, dtype=DTYPE_BF16):
        """nn_evals.evaluate (nn_evals.py:48-62): mean over the 8 dihedral images, one tower launch."""
        leaf_boards = np.ascontiguousarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
        n = len(leaf_boards)
        logits = np.zeros((n, 7, 7, 17), dtype=np.float32)
        values = np.zeros((n, 1), dtype=np.float32)
        check(load().azh_net_forward_sym(self.h, dtype, n, _ptr(leaf_boards), int(blockers), _ptr(logits), _ptr(values)))
        return logits, values

    def bench(self, n, iters=20, dtype=DTYPE_BF16, thin=False):
        """Average milliseconds per tower launch over n synthetic boards."""
        ms = ctypes.c_float(0)
        check((load().azh_net_bench_thin if thin else load().azh_net_bench)(self.h, dtype, n, iters, ctypes.byref(ms)))
        return float(ms.value)

    def stamps(self, n, wgs=64):
        out = np.zeros((wgs, 4, 128), dtype=np.uint64)
        check(load().azh_net_stamps(self.h, n, wgs, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            load().azh_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ engine

def run_engines(engines, net, iterations, dtype=DTYPE_BF16):
    """engine.run for several engines of one GPU, their iterations enqueued in turn (azh_engines_run)."""
    handles = (ctypes.c_void_p * len(engines))(*[e.h for e in engines])
    check(load().azh_engines_run(handles, len(engines), net.h, dtype, iterations))


class Engine:
    """Batched self-play search on the GPU (cpp/self_play_client.cpp's workers)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.G = cfg.games
        h = ctypes.c_void_p()
        check(load().azh_engine_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        self.node_cap = load().azh_engine_node_cap(h)
        self.edge_cap = load().azh_engine_edge_cap(h)
        self._json = np.zeros(1 << 22, dtype=np.uint8)

    def close(self):
        if getattr(self, "h", None):
            load().azh_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def select(self):
        n = ctypes.c_int32(0)
        check(load().azh_engine_select(self.h, ctypes.byref(n)))
        return int(n.value)

    def leaves(self):
        need = np.zeros(self.G, dtype=np.int32)
        boards = np.zeros((self.G, 2), dtype=np.uint64)
        check(load().azh_engine_leaves(self.h, _ptr(need), _ptr(boards)))
        return need, boards

    def leaf_features(self, n):
        out = np.zeros((max(n, 1), 7, 7, 4), dtype=np.float32)
        games = np.zeros(max(n, 1), dtype=np.int32)
        check(load().azh_engine_leaf_features(self.h, _ptr(out), _ptr(games)))
        return out[:n], games[:n]

    def eval(self, net, dtype=DTYPE_BF16):
        check(load().azh_engine_eval(self.h, net.h, dtype))

    def set_evals(self, logits, values):
        logits = np.ascontiguousarray(logits, dtype=np.float32).reshape(self.G, POLICY_SIZE)
        values = np.ascontiguousarray(values, dtype=np.float32).reshape(self.G)
        check(load().azh_engine_set_evals(self.h, _ptr(logits), _ptr(values)))

    def backup(self):
        check(load().azh_engine_backup(self.h))

    def run(self, net, iterations, dtype=DTYPE_BF16):
        check(load().azh_engine_run(self.h, net.h, dtype, iterations))

    def run_arena(self, net_a, net_b, iterations, dtype=DTYPE_BF16):
        check(load().azh_engine_run_arena(self.h, net_a.h, net_b.h, dtype, iterations))

    def sync(self):
        check(load().azh_engine_sync(self.h))

    def set_visits(self, visits):
        check(load().azh_engine_set_visits(self.h, visits))

    def set_thin_batches(self, mode):
        """0: the 3-board tower; 1: one board per workgroup (a handful of leaves per iteration); -1: by the engine's size."""
        check(load().azh_engine_set_thin_batches(self.h, int(mode)))

    def set_positions(self, boards, plies):
        """Every slot restarts at boards[g] (packed x | turn << 63, o) / plies[g] with a fresh tree (counted, not written)."""
        boards = np.ascontiguousarray(boards, dtype=np.uint64).reshape(self.G, 2)
        plies = np.ascontiguousarray(plies, dtype=np.int32).reshape(self.G)
        check(load().azh_engine_set_positions(self.h, _ptr(boards), _ptr(plies)))

    def set_emit_order(self, by_uid):
        """True: finished games are handed out in uid order (unbiased prefixes); False: as they finish."""
        check(load().azh_engine_set_emit_order(self.h, 1 if by_uid else 0))

    def set_game_limit(self, games):
        """Play uids 0 .. games - 1 only; slots past the limit go idle (call before the first iteration)."""
        check(load().azh_engine_set_game_limit(self.h, int(games)))

    def game_state(self, g):
        s = GameState()
        check(load().azh_engine_game_state(self.h, g, ctypes.byref(s)))
        return s

    def tree(self, g):
        s = self.game_state(g)
        boards = np.zeros((s.n_nodes, 2), dtype=np.uint64)
        info = np.zeros((s.n_nodes, 4), dtype=np.uint32)
        edges = np.zeros((s.n_edges, 4), dtype=np.uint32)
        moves = np.zeros(s.n_edges, dtype=np.uint16)
        check(load().azh_engine_tree(self.h, g, _ptr(boards), _ptr(info), _ptr(edges), _ptr(moves)))
        return boards, info, edges, moves

    def tree_raw(self, g):
        """The game's edges as the 16-byte device records (prior bits with the descent's mark in bit 31, score bits,
        visits | child << 16, the child's edge range): diagnostic, azh_engine_tree_raw."""
        edges = np.zeros((self.game_state(g).n_edges, 4), dtype=np.uint32)
        check(load().azh_engine_tree_raw(self.h, g, _ptr(edges)))
        return edges

    def stats(self):
        out = np.zeros(STAT_COUNT, dtype=np.uint64)
        check(load().azh_engine_stats(self.h, _ptr(out)))
        return {n: int(out[i]) for i, n in enumerate(STAT_NAMES)}

    def tree_stamps(self, net, dtype=DTYPE_BF16):
        """(G, 10): 8 s_memrealtime readings (100 MHz) of one stamped tree launch of the device loop (two iterations are
        run), then the levels descended and the children scanned."""
        out = np.zeros((self.G, 10), dtype=np.uint64)
        check(load().azh_engine_tree_stamps(self.h, net.h, dtype, _ptr(out)))
        return out

    def timing_reset(self, enable=True):
        """False / 0: off; True / 1: every iteration of the device loop is event-timed; n: every n-th."""
        check(load().azh_engine_timing_reset(self.h, int(enable)))

    def timing(self):
        t = Timing()
        check(load().azh_engine_timing(self.h, ctypes.byref(t)))
        return {"select_ms": t.select_ms, "net_ms": t.net_ms, "backup_ms": t.backup_ms, "iterations": t.iterations}

    def fetch(self):
        """Wait for the work enqueued so far and take its finished games off the device; the next drain_json formats
        them without touching the GPU (so the next run can be enqueued in between)."""
        check(load().azh_engine_fetch(self.h))

    def busy(self):
        """True while work enqueued on this engine's streams is still in flight (a query: never waits)."""
        rc = load().azh_engine_query(self.h)
        if rc < 0:
            check(rc)
        return rc == 1

    def implicit_fetches(self):
        """Times a drain had to fetch by itself (= waited for the device); 0 in a loop that fetches before it drains."""
        return int(load().azh_engine_implicit_fetches(self.h))

    def drain_json(self):
        """Finished games since the last call, as a list of JSON lines (bytes, no newline)."""
        lines = []
        while True:
            used = ctypes.c_int64(0)
            n = ctypes.c_int32(0)
            rc = load().azh_engine_drain_json(self.h, _ptr(self._json), self._json.nbytes, ctypes.byref(used),
                                              ctypes.byref(n))
            if rc == -6:
                self._json = np.zeros(self._json.nbytes * 2, dtype=np.uint8)
                continue
            check(rc)
            if n.value == 0:
                break
            lines.extend(bytes(self._json[:used.value]).split(b"\n")[:-1])
        return lines
