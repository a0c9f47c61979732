#!/usr/bin/env python3
"""Per-layer phase breakdown of the bf16 tower from in-kernel s_memtime stamps (diagnostic build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ataxxzero_amd import link, model
conv, bn = model.random_init(12, 128, seed=1)
net = link.Net(conv, bn)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = net.stamps(n, wgs=512).astype(np.int64)
st = st[(st[:, :, 0] != 0).all(axis=1)]
print("workgroups with stamps:", len(st))
t0, tprol, tend = st[:, :, 0], st[:, :, 1], st[:, :, 2]
print("prologue (zero LDS, planes) cycles: %.0f   whole kernel per wave: %.0f" % ((tprol - t0).mean(), (tend - t0).mean()))
L = np.stack([st[:, :, 8 + 4 * i: 12 + 4 * i] for i in range(24)], axis=2)  # [wg][wave][layer][4]
loop = L[..., 1] - L[..., 0]; epi = L[..., 2] - L[..., 1]; bar = L[..., 3] - L[..., 2]
nxt = L[:, :, 1:, 0] - L[:, :, :-1, 3]
print("per layer (mean over waves, layers 1..24): loop %.0f  epilogue %.0f  barrier-wait %.0f  next-layer prologue %.0f  cycles" % (
    loop.mean(), epi.mean(), bar.mean(), nxt.mean()))
print("conv1 (no skip) epilogue %.0f   conv2 (skip) epilogue %.0f" % (epi[:, :, 0::2].mean(), epi[:, :, 1::2].mean()))
print("ideal MFMA cycles per layer per wave: %d (x2 when two waves share a SIMD)" % (72 * (5 if os.environ.get("AZH_TOWER_BOARDS", "3") != "6" else 10) * 32))
l0 = st[:, :, 4:8]
print("layer0: loop %.0f epi %.0f bar %.0f" % ((l0[..., 1] - l0[..., 0]).mean(), (l0[..., 2] - l0[..., 1]).mean(), (l0[..., 3] - l0[..., 2]).mean()))
print("tail (heads + output) cycles: %.0f" % (tend - L[:, :, -1, 3]).mean())
