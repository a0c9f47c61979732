"""Host loop of the CPU baseline (tools/cpu_baseline/host_selfplay.hip): the reference's
accelerated_generate_games.py:54-83 loop — get_workload -> evaluate -> complete_workload — around
host game threads.  Used by bench.py's cpu_baseline leg only."""
import ctypes
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libazh_cpu_baseline.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB)
        L.cb_launch.restype = ctypes.c_int
        L.cb_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.cb_get_workload.restype = ctypes.c_int
        L.cb_get_workload.argtypes = [ctypes.c_void_p]
        L.cb_complete_workload.restype = None
        L.cb_complete_workload.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.cb_stats.restype = None
        L.cb_stats.argtypes = [ctypes.c_void_p]
        L.cb_shutdown.restype = None
        L.cb_shutdown.argtypes = []
        _lib = L
    return _lib


def stats():
    out = np.zeros(4, dtype=np.int64)
    lib().cb_stats(out.ctypes.data)
    return dict(zip(("steps", "evals", "plies", "games"), (int(v) for v in out)))


def run(evaluate, visits, buffer_entries, seconds, seed=1, warmup_seconds=1.0):
    """2 * buffer_entries game threads for `seconds` (after a warm-up); `evaluate(boards (B,2) u64)` ->
    (logits (B,833) f32, values (B,) f32) or None for the null evaluator (all-zero outputs, the reference's
    link.py:34-52 idea).  Returns steps/s etc. over the timed window."""
    L = lib()
    B = int(buffer_entries)
    boards = np.zeros((B, 2), dtype=np.uint64)
    zl, zv = np.zeros((B, 833), np.float32), np.zeros(B, np.float32)
    if L.cb_launch(int(visits), B, 2 * B, int(seed)) != 0:
        raise RuntimeError("cb_launch failed")
    try:
        t_start = time.perf_counter()
        s0, t0 = None, None
        batches = 0
        while True:
            now = time.perf_counter()
            if s0 is None and now - t_start >= warmup_seconds:
                s0, t0 = stats(), now
            if s0 is not None and now - t0 >= seconds:
                break
            w = L.cb_get_workload(boards.ctypes.data)
            if evaluate is None:
                logits, values = zl, zv
            else:
                logits, values = evaluate(boards)
                logits = np.ascontiguousarray(logits, dtype=np.float32).reshape(B, 833)
                values = np.ascontiguousarray(values, dtype=np.float32).reshape(B)
            L.cb_complete_workload(w, logits.ctypes.data, values.ctypes.data)
            batches += s0 is not None
        s1, t1 = stats(), time.perf_counter()
    finally:
        L.cb_shutdown()
    dt = t1 - t0
    return {"steps_per_s": (s1["steps"] - s0["steps"]) / dt, "evals_per_s": (s1["evals"] - s0["evals"]) / dt,
            "plies_per_s": (s1["plies"] - s0["plies"]) / dt, "seconds": dt, "batches": batches,
            "threads": 2 * B, "buffer_entries": B}
