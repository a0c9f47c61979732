#!/usr/bin/env python3
"""Would an fp8 tower pass the precision-in-the-loop gate?  A CPU study, run BEFORE any fp8 kernel is written.

The gate (round-2 review, item 9; tools/precision_in_the_loop.py measures it on the GPU for bf16 / f16): the same mid-game
positions are searched with the f32 net and with the reduced-precision net — same seeds, 400 sims, no Dirichlet noise, so
the evaluator is the only difference — and the most visited root move must agree in >= 98 % of the positions with a mean
total-variation distance of the root visit distributions <= 2 %.

Here the search is the CPU oracle's (step-wise API, one game slot per position) and the evaluator is the net in torch on
the CPU with the tower's rounding points emulated: batch-norm scale folded into the weights and the folded weights rounded;
every layer's output (after shift / residual add / relu — what the fused tower keeps in LDS) rounded; accumulation in f32.
Rows:
  f64            the search's own sensitivity: nothing but summation order / last-bit differences against f32
  bf16, f16      calibration against what the GPU measured with the real towers (profiles/round2_precision_in_the_loop.json,
                 random-init net: bf16 98.4 % / TV 1.7 %, f16 100 % / 0.02 %)
  fp8_tensor     e4m3 operands, one scale per layer for the activations (calibrated on the root positions), one per output
                 channel for the weights — what the unscaled fp8 MFMA needs
  fp8_mx         e4m3 operands with a power-of-two scale per 32 input channels, activations and weights — the block scaling
                 of v_mfma_scale_f32_16x16x128_f8f6f4
  *_res16        the same, but the residual stream stays in bf16 beside the fp8 operand copy (the LDS budget allows it)
In the fp8 rows the first layer (4 input planes) and the heads stay in bf16: 1.5 % of the flops.

This file lives under tests/ because it drives the oracle (test infrastructure); it is a study, not a test — pytest does
not collect it.

    python tests/studies/fp8_gate_emulation.py [--positions 128] [--visits 400] [--network trained.npy] > profiles/....txt
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from ataxxzero_amd import model  # noqa: E402
from oracle import net_oracle, oracle_lib as orc  # noqa: E402

E4M3_MAX = 448.0


def round_to(x, kind):
    if kind == "f32" or kind == "f64":
        return x
    if kind == "bf16":
        return x.to(torch.bfloat16).to(x.dtype)
    if kind == "f16":
        return x.to(torch.float16).to(x.dtype)
    raise ValueError(kind)


def e4m3(x):
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def fp8_blocks(x, dim, block=32):
    """e4m3 with one power-of-two scale per `block` consecutive elements along `dim` (OCP MX: the scale is the largest
    power of two that keeps the block's maximum inside the element format)."""
    x = x.movedim(dim, -1)
    shape = x.shape
    xb = x.reshape(shape[:-1] + (shape[-1] // block, block))
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / E4M3_MAX)))
    q = e4m3(xb / scale) * scale
    return q.reshape(shape).movedim(-1, dim)


class Tower:
    """model.py:38-79 with the fused tower's rounding points."""

    def __init__(self, conv, bn, mode):
        self.mode = mode
        self.dt = torch.float64 if mode == "f64" else torch.float32
        self.act16 = {"f64": "f64", "f32": "f32", "bf16": "bf16", "f16": "f16"}.get(mode, "bf16")
        self.fp8 = mode.startswith("fp8")
        self.mx = mode.startswith("fp8_mx")
        self.res16 = mode.endswith("_res16")
        blocks = (len(conv) - 5) // 2
        self.blocks = blocks
        self.w, self.shift = [], []
        for i in range(2 * blocks + 1):
            w = torch.tensor(np.asarray(conv[i], dtype=np.float64)).permute(3, 2, 0, 1)       # (o, c, kx, ky)
            scale = 1.0 / torch.sqrt(torch.tensor(np.asarray(bn[2 * i + 1], dtype=np.float64)) + net_oracle.BN_EPS)
            self.shift.append((-torch.tensor(np.asarray(bn[2 * i], dtype=np.float64)) * scale).to(self.dt).view(1, -1, 1, 1))
            w = (w * scale.view(-1, 1, 1, 1)).to(self.dt)
            self.w.append(self.quant_weight(w, first=(i == 0)))
        self.wp = round_to(torch.tensor(np.asarray(conv[2 * blocks + 1], dtype=np.float64)).permute(3, 2, 0, 1).to(self.dt), self.act16)
        self.wv = round_to(torch.tensor(np.asarray(conv[2 * blocks + 2], dtype=np.float64)).permute(3, 2, 0, 1).to(self.dt), self.act16)
        self.fc_w = torch.tensor(np.asarray(conv[2 * blocks + 3], dtype=np.float64)).to(self.dt)
        self.fc_b = torch.tensor(np.asarray(conv[2 * blocks + 4], dtype=np.float64)).to(self.dt)
        self.act_scale = None     # fp8_tensor: per-layer activation scales, set by calibrate()

    def quant_weight(self, w, first):
        if not self.fp8 or first:
            return round_to(w, self.act16).contiguous(memory_format=torch.channels_last)
        if self.mx:
            q = fp8_blocks(w, dim=1)                                       # 32 input channels of one (o, tap)
        else:
            s = w.abs().amax(dim=(1, 2, 3), keepdim=True) / E4M3_MAX       # one scale per output channel
            q = e4m3(w / s) * s
        return q.contiguous(memory_format=torch.channels_last)

    def operand(self, h, layer):
        """what the MFMA reads of the stored activation `h` as layer `layer`'s input"""
        if not self.fp8 or layer == 0:
            return h
        if self.mx:
            return fp8_blocks(h, dim=1)
        s = self.act_scale[layer]
        return e4m3(h / s) * s

    def store(self, y):
        """what the tower keeps of a layer's output"""
        if self.fp8 and not self.res16:
            return y          # kept exactly as the next operand() rounds it: the rounding happens there, once
        return round_to(y, self.act16)

    def conv(self, h, layer, record=None):
        x = self.operand(h, layer)
        if record is not None:
            record[layer] = max(record.get(layer, 0.0), float(h.abs().max()))
        return F.conv2d(x, self.w[layer], padding=1) + self.shift[layer]

    def forward(self, feats, record=None):
        h = torch.tensor(np.asarray(feats, dtype=np.float64)).permute(0, 3, 1, 2).to(self.dt)
        h = h.contiguous(memory_format=torch.channels_last)
        h = self.store(torch.relu(self.conv(h, 0, record)))
        if self.fp8 and not self.res16:
            h = self.operand(h, 1)                       # stored as fp8: the residual stream carries the rounded value
        for b in range(self.blocks):
            i1, i2 = 1 + 2 * b, 2 + 2 * b
            t = self.store(torch.relu(self.conv(h, i1, record)))
            t = self.conv(t, i2, record)
            h = self.store(torch.relu(t + h))
            if self.fp8 and not self.res16:
                h = self.operand(h, min(i2 + 1, 2 * self.blocks))
        hq = round_to(h, self.act16)                     # heads in 16 bit
        policy = F.conv2d(hq, self.wp).permute(0, 2, 3, 1).reshape(len(h), -1)
        v = F.conv2d(hq, self.wv).permute(0, 2, 3, 1).reshape(len(h), 49)
        value = torch.tanh(v @ self.fc_w + self.fc_b).reshape(-1)
        return policy.to(torch.float32).numpy(), value.to(torch.float32).numpy()

    def calibrate(self, feats, f32_tower):
        """per-layer activation scales of fp8_tensor: the largest stored activation the f32 net shows on `feats`, mapped
        to the top of the e4m3 range"""
        rec = {}
        f32_tower.forward(feats, rec)
        self.act_scale = {k: v / E4M3_MAX for k, v in rec.items()}
        self.act_scale[2 * self.blocks] = self.act_scale.get(2 * self.blocks, 1.0 / E4M3_MAX)


def midgame_positions(n, seed, blockers_fen):
    """(x | turn << 63, o) pairs from uniformly random play (oracle rules), ply 8 .. 60"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        p = orc.pos_from_fen(blockers_fen)
        target = int(rng.integers(8, 61))
        ok = True
        for _ in range(target):
            if orc.result(p) != 0:
                ok = False
                break
            mv = orc.movegen(p)
            if len(mv) == 0:
                orc.lib().orc_pass(p)
                continue
            m = int(mv[int(rng.integers(0, len(mv)))])
            orc.lib().orc_makemove(p, m & 0xFF, m >> 8)
        if ok and orc.result(p) == 0 and len(orc.movegen(p)) > 1:
            out.append((int(p.pieces[0]) | (int(p.turn) << 63), int(p.pieces[1])))
    return np.array(out, dtype=np.uint64)


def search(tower, boards, blockers, visits, seed):
    n = len(boards)
    cfg = orc.make_config(n, visits, seed=seed, weight=0.0)
    e = orc.Engine(cfg)
    e.set_positions(boards, np.zeros(n, dtype=np.int32))
    for _ in range(visits + 1):                           # one root evaluation + `visits` steps
        e.select()
        feats = net_oracle.features_from_leaf_boards(e.leaf_boards(), blockers, dtype=np.float32)
        logits, values = tower.forward(feats)
        e.backup(logits, values)
    out = []
    for g in range(n):
        s = e.game_state(g)
        assert s.ply == 0 and s.root_visits == visits, s.as_tuple()
        _, info, edges, moves = e.tree(g)
        first, m = int(info[0, 0]), int(info[0, 1] & 0xFFFF)
        out.append((moves[first:first + m].copy(), edges[first:first + m, 1].astype(np.float64)))   # rows: prior, visits, score, child
    e.close()
    return out


def compare(ref, other):
    top1, kls, tvs = [], [], []
    for (mv_a, n_a), (mv_b, n_b) in zip(ref, other):
        assert (mv_a == mv_b).all()
        p, q = n_a / n_a.sum(), n_b / n_b.sum()
        top1.append(int(np.argmax(n_a) == np.argmax(n_b)))
        eps = 0.5 / n_a.sum()
        kls.append(float(np.sum(p * np.log((p + eps) / (q + eps)))))
        tvs.append(float(0.5 * np.abs(p - q).sum()))
    return {"top1_agreement": float(np.mean(top1)), "tv_mean": float(np.mean(tvs)), "kl_mean": float(np.mean(kls)),
            "kl_p95": float(np.percentile(kls, 95)), "kl_max": float(np.max(kls))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=128)
    ap.add_argument("--visits", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--network", help=".npy weights; default: the random-init net of the GPU measurement (seed 1)")
    ap.add_argument("--modes", default="f64,bf16,f16,fp8_tensor,fp8_mx,fp8_tensor_res16,fp8_mx_res16")
    args = ap.parse_args()
    torch.manual_seed(0)
    conv, bn = model.load_model(args.network) if args.network else model.random_init(args.blocks, 128, seed=1)
    blockers = int(orc.pos_from_fen(orc.START_FEN_SELFPLAY).blockers)
    boards = midgame_positions(args.positions, args.seed, orc.START_FEN_SELFPLAY)
    low = (1 << 63) - 1
    roots = np.array([[int(x) & low, int(o)] if int(x) >> 63 == 0 else [int(o), int(x) & low] for x, o in boards],
                     dtype=np.uint64)                     # (mover, opponent), as the engine hands leaves to the net
    root_feats = net_oracle.features_from_leaf_boards(roots, blockers, dtype=np.float32)
    f32 = Tower(conv, bn, "f32")
    p32, v32 = f32.forward(root_feats)
    t0 = time.time()
    ref = search(f32, boards, blockers, args.visits, args.seed)
    report = {"positions": int(len(boards)), "visits": args.visits,
              "net": os.path.basename(args.network) if args.network else "%dx128 random-init seed 1" % args.blocks,
              "logit_scale": float(np.abs(p32).max()), "gate": {"top1_agreement": ">= 0.98", "tv_mean": "<= 0.02"},
              "f32_search_seconds": round(time.time() - t0, 1)}
    print(json.dumps(report), flush=True)
    for mode in args.modes.split(","):
        tw = Tower(conv, bn, mode)
        if mode.startswith("fp8_tensor"):
            tw.calibrate(root_feats, f32)
        p, v = tw.forward(root_feats)
        r = compare(ref, search(tw, boards, blockers, args.visits, args.seed))
        r.update(max_abs_dlogit=float(np.abs(p - p32).max()), mean_abs_dlogit=float(np.abs(p - p32).mean()),
                 max_abs_dvalue=float(np.abs(v - v32).max()))
        r["passes_gate"] = bool(r["top1_agreement"] >= 0.98 and r["tv_mean"] <= 0.02)
        print(json.dumps({mode: r}), flush=True)


if __name__ == "__main__":
    main()
