#!/usr/bin/env python3
"""PCIe-inclusive rate of the compatibility path: the reference's four-symbol ABI hands host buffers over every
iteration ((B,7,7,4) f32 features out, (B,7,7,17) f32 logits + (B,1) values in).  Null evaluator (zero logits), so the
figure is the hand-off + tree side alone; a real host evaluator adds its own time."""
import ctypes, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ataxxzero_amd import link
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
visits = 400
out = os.path.join(tempfile.mkdtemp(), "games.json")
bufs = [np.zeros((B, 7, 7, 4), dtype=np.float32) for _ in (0, 1)]
p = np.zeros((B, 7, 7, 17), dtype=np.float32)
v = np.zeros((B, 1), dtype=np.float32)
link.launch_threads(out.encode(), visits, ctypes.c_void_p(bufs[0].ctypes.data), ctypes.c_void_p(bufs[1].ctypes.data), B, 2 * B)
try:
    for _ in range(200):
        w = link.get_workload()
        link.complete_workload(w, ctypes.c_void_p(p.ctypes.data), ctypes.c_void_p(v.ctypes.data))
    rows = 0
    t0 = time.perf_counter()
    n = 600
    for _ in range(n):
        w = link.get_workload()
        rows += int((bufs[w][:, 0, 0, 0] == 1).sum())
        link.complete_workload(w, ctypes.c_void_p(p.ctypes.data), ctypes.c_void_p(v.ctypes.data))
    dt = time.perf_counter() - t0
finally:
    link.shutdown()
bytes_per = B * (196 + 833 + 1) * 4
print("reference ABI, %d game slots, buffer %d rows: %.2f ms per workload, %.0f leaf rows/s, %.2f GB/s over PCIe (features + logits + values)" % (
    2 * B, B, 1e3 * dt / n, rows / dt, bytes_per * n / dt / 1e9))
