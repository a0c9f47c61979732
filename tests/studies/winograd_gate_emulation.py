#!/usr/bin/env python3
"""Would a Winograd F(2x2, 3x3) tower pass the precision-in-the-loop gate?  A CPU study, run BEFORE any kernel is written
(round-3 review, item 4 — priced the way the fp8 tower was, tests/studies/fp8_gate_emulation.py, whose search, positions,
comparison and calibration rows this script reuses).

The arithmetic reduction: a 3x3 convolution over the 8x8-padded board as 16 tiles of 2x2 outputs per board; per tile and
channel the 4x4 input patch d becomes V = B^T d B, the 3x3 filter g becomes U = G g G^T (offline), the 16 products U (.) V
are summed over the input channels — sixteen 128 x 128 GEMMs over (board, tile) columns instead of nine over cells — and
Y = A^T M A gives the 2x2 outputs: 16 multiplies per 4 outputs instead of 36 (2.25x), 1.56x fewer MFMAs per layer and
workgroup once the tower's tile quantisation and its skipped border taps are counted (1,536 against about 2,400).

Rounding points emulated (everything else as in the direct tower of fp8_gate_emulation.Tower):
  U   computed in float64 from the batch-norm-folded filter, rounded to the operand type (G holds halves: U's entries span a
      wider range than g's)
  V   B^T d B of the stored (already rounded) activations; entries are sums of up to four activations.
        wino_*     the sums in f32, one rounding to the operand type
        wino_*_pk  the sums in packed 16-bit arithmetic (v_pk_add_f16 / bf16: what keeps the transform's VALU cost down), a
                   rounding after each of the two passes
  M   accumulated in f32 (the MFMA), Y = A^T M A in f32, then shift / residual / relu and the stored rounding as before
The first layer (4 input planes) and the heads stay direct.

Gate (the same as for fp8): most visited root move equal to the f32 search's in >= 98 % of the positions, mean total
variation of the root visit distributions <= 2 %; asked for f16 Winograd on a TRAINED net.

    python tests/studies/winograd_gate_emulation.py [--network trained.npy] [--positions 128] [--visits 400]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import fp8_gate_emulation as base  # noqa: E402
from ataxxzero_amd import model  # noqa: E402
from oracle import net_oracle, oracle_lib as orc  # noqa: E402

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


class WinogradTower(base.Tower):
    def __init__(self, conv, bn, mode):
        kind = mode.split("_")[1]                       # wino_f16, wino_bf16, wino_f16_pk, wino_f32
        super().__init__(conv, bn, kind)
        self.kind = kind
        self.pk = mode.endswith("_pk")
        self.U = {}
        for i in range(1, 2 * self.blocks + 1):
            # folded filter in float64 (as the direct tower folds it), transformed, then rounded once
            w = torch.tensor(np.asarray(conv[i], dtype=np.float64)).permute(3, 2, 0, 1)
            scale = 1.0 / torch.sqrt(torch.tensor(np.asarray(bn[2 * i + 1], dtype=np.float64)) + net_oracle.BN_EPS)
            w = w * scale.view(-1, 1, 1, 1)
            u = torch.einsum("ai,ocij,bj->aboc", G, w, G)                   # (4, 4, o, c)
            self.U[i] = base.round_to(u.to(self.dt), kind).reshape(16, u.shape[2], u.shape[3]).contiguous()
        self.bt = BT.to(self.dt)
        self.at = AT.to(self.dt)

    def conv(self, h, layer, record=None):
        if layer == 0:
            return super().conv(h, layer, record)
        n, c = h.shape[0], h.shape[1]
        xp = torch.nn.functional.pad(h.contiguous(), (1, 2, 1, 2))               # 10 x 10: one ring + the 8th row / column
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # (n, c, 4, 4, 4, 4): tile y, tile x, i, j
        if self.pk:
            v = base.round_to(torch.einsum("ai,nctuij->nctuaj", self.bt, d), self.kind)
            v = base.round_to(torch.einsum("nctuaj,bj->nctuab", v, self.bt), self.kind)
        else:
            v = base.round_to(torch.einsum("ai,nctuij,bj->nctuab", self.bt, d, self.bt), self.kind)
        v = v.permute(4, 5, 1, 0, 2, 3).reshape(16, c, n * 16)                   # (position, c, board x tile)
        m = torch.bmm(self.U[layer], v)                                          # (16, o, n * 16), f32 accumulation
        o = m.shape[1]
        m = m.reshape(4, 4, o, n, 4, 4)
        y = torch.einsum("ia,abontu,jb->notiuj", self.at, m, self.at)            # (n, o, tile y, i, tile x, j)
        y = y.reshape(n, o, 8, 8)[:, :, :7, :7]
        return y + self.shift[layer]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=128)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--network", help=".npy weights; default: the random-init 12x128 net (seed 1)")
    ap.add_argument("--modes", default="wino_f32,f16,wino_f16,wino_f16_pk,bf16,wino_bf16,wino_bf16_pk")
    args = ap.parse_args()
    torch.manual_seed(0)
    conv, bn = model.load_model(args.network) if args.network else model.random_init(args.blocks, 128, seed=1)
    blockers = int(orc.pos_from_fen(orc.START_FEN_SELFPLAY).blockers)
    boards = base.midgame_positions(args.positions, args.seed, orc.START_FEN_SELFPLAY)
    low = (1 << 63) - 1
    roots = np.array([[int(x) & low, int(o)] if int(x) >> 63 == 0 else [int(o), int(x) & low] for x, o in boards],
                     dtype=np.uint64)
    root_feats = net_oracle.features_from_leaf_boards(roots, blockers, dtype=np.float32)
    f32 = base.Tower(conv, bn, "f32")
    p32, v32 = f32.forward(root_feats)
    t0 = time.time()
    ref = base.search(f32, boards, blockers, args.visits, args.seed)
    print(json.dumps({"positions": int(len(boards)), "visits": args.visits,
                      "net": os.path.basename(args.network) if args.network else "%dx128 random-init seed 1" % args.blocks,
                      "logit_scale": float(np.abs(p32).max()), "gate": {"top1_agreement": ">= 0.98", "tv_mean": "<= 0.02"},
                      "f32_search_seconds": round(time.time() - t0, 1)}), flush=True)
    for mode in args.modes.split(","):
        tw = WinogradTower(conv, bn, mode) if mode.startswith("wino_") else base.Tower(conv, bn, mode)
        p, v = tw.forward(root_feats)
        t0 = time.time()
        r = base.compare(ref, base.search(tw, boards, blockers, args.visits, args.seed))
        r.update(max_abs_dlogit=float(np.abs(p - p32).max()), mean_abs_dlogit=float(np.abs(p - p32).mean()),
                 max_abs_dvalue=float(np.abs(v - v32).max()), seconds=round(time.time() - t0, 1))
        r["passes_gate"] = bool(r["top1_agreement"] >= 0.98 and r["tv_mean"] <= 0.02)
        print(json.dumps({mode: r}), flush=True)


if __name__ == "__main__":
    main()
