#!/usr/bin/env python3
"""Net-only timing of the fused tower kernel (HIP events inside the library).
usage: python tools/net_bench.py [n ...]    env AZH_TOWER_BOARDS=3|6 picks the tile variant, THIN=1 the one-board-per-workgroup kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ataxxzero_amd import link, model  # noqa: E402

blocks = int(os.environ.get("BLOCKS", "12"))
conv, bn = model.random_init(blocks, 128, seed=1)
net = link.Net(conv, bn)
flops = model.flops_per_eval(blocks, 128)
for dt_name in os.environ.get("DTYPES", "bf16").split(","):
    dt = link.DTYPES[dt_name]
    for n in [int(a) for a in sys.argv[1:]] or [4096, 16384]:
        thin = bool(os.environ.get("THIN"))     # THIN=1: one board per workgroup (azh_net_bench_thin)
        ms = net.bench(n, iters=10, dtype=dt, thin=thin)
        print("boards/wg=%s dtype=%s n=%d: %.3f ms/launch  %.2f M evals/s  %.1f TFLOP/s (%.1f%% of %s peak)" % (
            "1" if thin else os.environ.get("AZH_TOWER_BOARDS", "3"), dt_name, n, ms, n / ms / 1e3, n * flops / ms / 1e9,
            100 * n * flops / ms / 1e9 / (157.3 if dt_name == "f32" else 2500.0), dt_name))
