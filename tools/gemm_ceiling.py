#!/usr/bin/env python3
"""What a dense bf16 GEMM reaches on this box through the vendor library (torch.matmul -> hipBLASLt/rocBLAS), on random
data, beside the fused tower's rate: the practical MFMA ceiling of the chip under its power limit (the 2.5 PFLOP/s
peak assumes 2.4 GHz; MFMA-dense kernels hold 1.8-1.9 GHz).  Prints TFLOP/s for a few large shapes, then the tower."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gemm(m, n, k, dtype, iters=30):
    import torch
    a = torch.randn(m, k, device="cuda", dtype=dtype)
    b = torch.randn(k, n, device="cuda", dtype=dtype)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return 2.0 * m * n * k / dt / 1e12


def main():
    import torch
    if "--json" in sys.argv:   # bench.py's context line: two large shapes, one JSON object
        import json
        shapes = ((8192, 8192, 8192), (4096, 4096, 16384))
        print(json.dumps({"unit": "TFLOP/s", "what": "torch.matmul bf16, random data, best of %s" % (shapes,),
                          "value": max(gemm(m, n, k, torch.bfloat16, iters=20) for m, n, k in shapes)}))
        return
    best = 0.0
    for shape in ((8192, 8192, 8192), (16384, 8192, 4096), (4096, 4096, 16384), (16384, 16384, 2048),
                  # the tower's own GEMM per layer at 16 K boards: M = cells, N = 128 channels, K = 9 * 128
                  (16384 * 49, 128, 1152)):
        tf = gemm(*shape, torch.bfloat16)
        best = max(best, tf)
        print("bf16 GEMM %7d x %5d x %5d: %7.1f TFLOP/s (%.1f%% of 2500)" % (shape + (tf, 100 * tf / 2500)))
    from ataxxzero_amd import link, model
    conv, bn = model.random_init(12, 128, seed=1)
    net = link.Net(conv, bn)
    ms = net.bench(16384, iters=10, dtype=link.DTYPE_BF16)
    tower = 16384 * model.flops_per_eval(12, 128) / ms / 1e9
    print("fused tower, 16384 boards: %.1f TFLOP/s (%.1f%% of 2500, %.0f%% of the best library GEMM above)" % (
        tower, 100 * tower / 2500, 100 * tower / best))


if __name__ == "__main__":
    main()
