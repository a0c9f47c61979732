"""Training side of the loop on PyTorch-ROCm (reference train.py + model.py's training graph).

Sample pipeline restated from train.py:11-89 (random game, random ply, side to move from ply
parity, value target +-1, one of the 8 dihedral symmetries, policy target from `dists` or a
one-hot of the move played), drawing from Python's `random` in the reference's order so the
same games file gives the same minibatches.  Loss and optimiser from model.py:81-101: softmax
cross-entropy on the 833 move logits + value MSE + 1e-4 * l2_loss over the trainable variables,
Momentum(0.9).  The trained net is written in the reference's `.npy` layout (model.py:179-183).

Q1: the reference trains batch-norm gamma/beta but never saves them, so
the net it plays with is not the net it trained.  Here gamma = 1, beta = 0 are fixed
(`affine=False`) by default — what is saved is exactly what was trained; `reference_bn_affine=True`
restores the reference's train-then-drop behaviour.
"""
import json
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import model

BOARD = model.BOARD_SIZE
MOVE_TYPES = model.MOVE_TYPES
# engine.py:75: layer of a jump = rank of (dx, dy) among the distance-2 offsets (ataxx_rules.py:17-20)
FAR_OFFSETS = [(a, b) for a in (-2, -1, 0, 1, 2) for b in (-2, -1, 0, 1, 2)
               if (a, b) != (0, 0) and (a, b) not in [(i, j) for i in (-1, 0, 1) for j in (-1, 0, 1)]]
DELTA_LAYER = {d: i for i, d in enumerate(FAR_OFFSETS)}


# ---------------------------------------------------------------- sample pipeline (train.py:11-89)

def uai_decode_move(text):
    """UAI text -> the reference's move value ("pass" | ("c", (x, y)) | ((x0, y0), (x1, y1)))."""
    from .uai import text_to_xy_move
    return text_to_xy_move(text)


def board_to_features(cells, to_move):
    """Four planes [x][y][c] — ones, mover's stones, opponent's stones, blockers (none in training files) — from the 49
    recorded cells (index x + 7 y), engine.py:53-73."""
    grid = np.asarray(cells, dtype=np.int8).reshape(BOARD, BOARD).T      # grid[x, y]
    planes = np.zeros((BOARD, BOARD, 4), dtype=np.int8)
    planes[..., 0] = 1
    planes[..., 1] = grid == to_move
    planes[..., 2] = (grid != 0) & (grid != to_move)
    return planes


def _symmetry_bits(index):
    if not 0 <= index < 8:
        raise ValueError("symmetry index %r" % (index,))
    return bool(index & 1), bool(index & 2), bool(index & 4)


def apply_symmetry(index, arr):
    """The dihedral symmetry `index` of train.py:11-23 on an [x][y][c] array: bit 0 mirrors x, bit 1 mirrors y,
    bit 2 then transposes; always a fresh array."""
    flip_x, flip_y, transpose = _symmetry_bits(index)
    out = np.array(arr)
    if flip_x:
        out = np.flip(out, axis=0)
    if flip_y:
        out = np.flip(out, axis=1)
    if transpose:
        out = np.transpose(out, (1, 0, 2))
    return np.ascontiguousarray(out)


def apply_symmetry_to_move(index, move):
    """The same symmetry on a move value (train.py:25-40)."""
    flip_x, flip_y, transpose = _symmetry_bits(index)
    last = BOARD - 1

    def image(xy):
        x = last - xy[0] if flip_x else xy[0]
        y = last - xy[1] if flip_y else xy[1]
        return (y, x) if transpose else (x, y)

    return tuple(part if part == "c" else image(part) for part in move)


def add_move_to_heatmap(heatmap, move, coef=1):
    """Policy plane of a move: clones in the last layer of the destination cell, jumps in the layer of their
    (dx, dy) (engine.py:75-87)."""
    source, (tx, ty) = move
    layer = MOVE_TYPES - 1 if source == "c" else DELTA_LAYER[(tx - source[0], ty - source[1])]
    heatmap[tx, ty, layer] += coef


def _policy_target(entry, ply, symmetry_index):
    target = np.zeros((BOARD, BOARD, MOVE_TYPES), dtype=np.float32)
    if "dists" in entry:
        weighted = entry["dists"][ply].items()
    else:
        weighted = [(entry["moves"][ply], 1)]       # one-hot on the move played (random / teacher games)
    for mv, weight in weighted:
        if isinstance(mv, str):
            mv = uai_decode_move(mv)
        else:                                       # json turned the tuples into lists
            mv = (mv[0] if mv[0] == "c" else tuple(mv[0]), tuple(mv[1]))
        add_move_to_heatmap(target, apply_symmetry_to_move(symmetry_index, mv), weight)
    if abs(1 - target.sum()) >= 1e-3:
        raise AssertionError("policy target does not sum to one")
    return target


def get_sample_from_entries(entries):
    """One training sample (train.py:43-77).  The draws from `random` come in the reference's order — game, ply,
    symmetry — so a games file yields the same minibatches as there."""
    while True:
        entry = random.choice(entries)
        ply = random.randrange(len(entry["boards"]))
        if "random_ply" in entry:                   # ONE_RANDOM_MOVE games: the position right after the random move
            ply = entry["random_ply"] + 1
        if entry["moves"][ply] == "pass":
            continue
        mover = 1 + ply % 2                         # x (1) moves on even plies
        symmetry_index = random.randrange(8)
        features = apply_symmetry(symmetry_index, board_to_features(entry["boards"][ply], mover))
        value = [1 if entry["result"] == mover else -1]
        return features, _policy_target(entry, ply, symmetry_index), value


def load_entries(paths):
    entries = []                                                          # train.py:79-89
    for path in paths:
        with open(path) as f:
            for line in f:
                line = line.strip()
                if line:
                    entries.append(json.loads(line))
    random.shuffle(entries)
    return entries


def make_minibatch_reference(entries, size):
    """`size` samples drawn one by one with get_sample_from_entries (train.py:123-130)."""
    feats, pols, vals = [], [], []
    for _ in range(size):
        f, p, v = get_sample_from_entries(entries)
        feats.append(f)
        pols.append(p)
        vals.append(v)
    return (np.asarray(feats, dtype=np.float32), np.asarray(pols, dtype=np.float32), np.asarray(vals, dtype=np.float32))


def _symmetry_tables():
    """For each of the 8 symmetries: which recorded cell (index x + 7 y) every feature cell [x][y] shows, and where every
    flat policy index 119 x + 17 y + layer goes — both read off apply_symmetry / apply_symmetry_to_move themselves."""
    cell_of = np.arange(BOARD * BOARD, dtype=np.int64).reshape(BOARD, BOARD).T[..., None]      # [x][y] -> x + 7 y
    cell_src = np.stack([apply_symmetry(s, cell_of)[..., 0].reshape(-1) for s in range(8)])       # [s][x * 7 + y]
    policy_to = np.zeros((8, BOARD * BOARD * MOVE_TYPES), dtype=np.int64)
    probe = np.zeros((BOARD, BOARD, MOVE_TYPES))

    def flat(move):
        probe[...] = 0
        add_move_to_heatmap(probe, move)
        return int(np.flatnonzero(probe.ravel())[0])

    for x in range(BOARD):
        for y in range(BOARD):
            moves = [("c", (x, y))] + [((x - dx, y - dy), (x, y)) for dx, dy in FAR_OFFSETS
                                       if 0 <= x - dx < BOARD and 0 <= y - dy < BOARD]
            for m in moves:
                for s in range(8):
                    policy_to[s, flat(m)] = flat(apply_symmetry_to_move(s, m))
    return cell_src, policy_to


_TABLES = None
# Per-entry arrays of the batched pipeline, kept OUTSIDE the entries (they stay plain JSON data) and bounded: keyed by
# id(entry) with the entry itself held, so an id is never reused while its row is alive; oldest rows go first.
_ARRAYS = {}
_ARRAYS_MAX = 100000      # entries; about 15 KB each (int8 cells, int16 indices, float32 weights)


_FLAT_INDEX = {}          # move as the entry spells it (UAI text, or json's nested lists as tuples) -> flat policy index


def _flat_policy_index(mv):
    """Flat index 119 x + 17 y + layer of a move's cell in the policy heat map, through add_move_to_heatmap itself, once per
    distinct move (there are 529): a games file names the same few hundred moves millions of times."""
    key = mv if isinstance(mv, str) else (mv[0] if mv[0] == "c" else tuple(mv[0]), tuple(mv[1]))
    flat = _FLAT_INDEX.get(key)
    if flat is None:
        probe = np.zeros((BOARD, BOARD, MOVE_TYPES))
        add_move_to_heatmap(probe, uai_decode_move(key) if isinstance(key, str) else key)
        flat = _FLAT_INDEX[key] = int(np.flatnonzero(probe.ravel())[0])
    return flat


def _entry_arrays(entry):
    """Per-entry arrays for the batched pipeline, built on first use: cells [plies][49] and, per ply, the flat policy
    indices and weights of its target (the visit distribution, or the move played)."""
    row = _ARRAYS.get(id(entry))
    cache = row[1] if row is not None and row[0] is entry else None
    if cache is None:
        cells = np.asarray(entry["boards"], dtype=np.int8)
        idx, wts, start = [], [], [0]
        for ply in range(len(entry["boards"])):
            if "dists" in entry:
                items = entry["dists"][ply].items()
            else:
                items = [(entry["moves"][ply], 1)]
            for mv, w in items:
                if mv == "pass":
                    continue
                idx.append(_flat_policy_index(mv))
                wts.append(w)
            start.append(len(idx))
        cache = (cells, np.asarray(idx, dtype=np.int16), np.asarray(wts, dtype=np.float32),
                 np.asarray(start, dtype=np.int32))
        if len(_ARRAYS) >= _ARRAYS_MAX:
            for key in list(_ARRAYS)[:_ARRAYS_MAX // 10]:
                del _ARRAYS[key]
        _ARRAYS[id(entry)] = (entry, cache)
    return cache


def make_minibatch(entries, size):
    """The same `size` samples as make_minibatch_reference — same draws from `random` in the same order, bit-identical
    arrays (tests/test_training.py) — assembled with array operations instead of per-sample Python: the sample pipeline,
    not the GPU step, bounded train.py's rate."""
    global _TABLES
    if _TABLES is None:
        _TABLES = _symmetry_tables()
    cell_src, policy_to = _TABLES
    cells, movers, syms, results, rows, pidx, pw = [], [], [], [], [], [], []
    while len(cells) < size:
        entry = random.choice(entries)                      # the reference's draws, in its order (train.py:45-60)
        ply = random.randrange(len(entry["boards"]))
        if "random_ply" in entry:
            ply = entry["random_ply"] + 1
        if entry["moves"][ply] == "pass":
            continue
        sym = random.randrange(8)
        c, idx, wts, start = _entry_arrays(entry)
        lo, hi = start[ply], start[ply + 1]
        rows.append(np.full(hi - lo, len(cells), dtype=np.int64))
        pidx.append(policy_to[sym, idx[lo:hi]])
        pw.append(wts[lo:hi])
        cells.append(c[ply])
        movers.append(1 + ply % 2)
        syms.append(sym)
        results.append(entry["result"])
    cells = np.stack(cells)
    movers = np.asarray(movers, dtype=np.int8)[:, None]
    shown = np.take_along_axis(cells, cell_src[np.asarray(syms)], axis=1)        # [b][x * 7 + y]
    feats = np.zeros((size, BOARD * BOARD, 4), dtype=np.float32)
    feats[..., 0] = 1
    feats[..., 1] = shown == movers
    feats[..., 2] = (shown != 0) & (shown != movers)
    pols = np.zeros((size, BOARD * BOARD * MOVE_TYPES), dtype=np.float32)
    np.add.at(pols, (np.concatenate(rows), np.concatenate(pidx)), np.concatenate(pw))
    if (np.abs(1 - pols.sum(axis=1)) >= 1e-3).any():
        raise AssertionError("policy target does not sum to one")
    vals = np.where(np.asarray(results)[:, None] == movers, 1, -1).astype(np.float32)
    return feats.reshape(size, BOARD, BOARD, 4), pols.reshape(size, BOARD, BOARD, MOVE_TYPES), vals


# ---------------------------------------------------------------- the network (model.py:38-101)

class Network(nn.Module):
    """model.Network in NCHW: tensors are [n][c][x][y] (the reference's [n][x][y][c] permuted)."""

    def __init__(self, blocks=12, filters=128, reference_bn_affine=False):
        super().__init__()
        self.blocks, self.filters = blocks, filters

        def conv(cin, cout, k):
            return nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)

        def bn():
            # tf.layers.batch_normalization defaults: momentum 0.99 (torch: 1 - 0.99), epsilon 1e-3
            return nn.BatchNorm2d(filters, eps=model.BN_EPSILON, momentum=0.01, affine=reference_bn_affine)

        self.convs = nn.ModuleList([conv(4, filters, 3)] + [conv(filters, filters, 3) for _ in range(2 * blocks)])
        self.bns = nn.ModuleList([bn() for _ in range(2 * blocks + 1)])
        self.policy = conv(filters, MOVE_TYPES, 1)
        self.value_conv = conv(filters, 1, 1)
        self.fc = nn.Linear(BOARD * BOARD, 1)

    def forward(self, x):
        h = F.relu(self.bns[0](self.convs[0](x)))
        for b in range(self.blocks):
            t = F.relu(self.bns[1 + 2 * b](self.convs[1 + 2 * b](h)))
            t = self.bns[2 + 2 * b](self.convs[2 + 2 * b](t))
            h = F.relu(t + h)
        policy = self.policy(h)                                            # [n][17][x][y]
        v = self.value_conv(h).reshape(x.shape[0], BOARD * BOARD)          # index x*7 + y, as tf.reshape of [n,x,y,1]
        value = torch.tanh(self.fc(v))
        return policy.permute(0, 2, 3, 1), value                           # policy back to [n][x][y][17]

    # .npy layout <-> torch: conv HWIO (i, j, c, o) <-> OIHW with H = x (i), W = y (j)
    def load_numpy(self, conv_weights, bn_params):
        with torch.no_grad():
            convs = list(self.convs) + [self.policy, self.value_conv]
            for m, w in zip(convs, conv_weights[:len(convs)]):
                m.weight.copy_(torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1)))))
            self.fc.weight.copy_(torch.from_numpy(np.asarray(conv_weights[-2]).reshape(1, BOARD * BOARD)))
            self.fc.bias.copy_(torch.from_numpy(np.asarray(conv_weights[-1]).reshape(1)))
            for i, m in enumerate(self.bns):
                m.running_mean.copy_(torch.from_numpy(np.asarray(bn_params[2 * i])))
                m.running_var.copy_(torch.from_numpy(np.asarray(bn_params[2 * i + 1])))

    def to_numpy(self):
        convs = list(self.convs) + [self.policy, self.value_conv]
        cw = [np.ascontiguousarray(np.transpose(m.weight.detach().cpu().numpy(), (2, 3, 1, 0))) for m in convs]
        cw.append(self.fc.weight.detach().cpu().numpy().reshape(BOARD * BOARD, 1).copy())
        cw.append(self.fc.bias.detach().cpu().numpy().reshape(1).copy())
        bn = []
        for m in self.bns:
            bn.append(m.running_mean.detach().cpu().numpy().copy())
            bn.append(m.running_var.detach().cpu().numpy().copy())
        return cw, bn


def losses(net, features, policies, values):
    """model.py:81-93: (policy cross-entropy, value MSE, 1e-4 * sum(l2_loss(w)) over trainable variables)."""
    x = features.permute(0, 3, 1, 2)
    logits, v = net(x)
    n = features.shape[0]
    logp = F.log_softmax(logits.reshape(n, -1), dim=1)
    policy_loss = -(policies.reshape(n, -1) * logp).sum(dim=1).mean()
    value_loss = ((values - v) ** 2).mean()
    reg = 1e-4 * sum((p ** 2).sum() / 2 for p in net.parameters() if p.requires_grad)
    return policy_loss, value_loss, reg


def train(games_paths, old_path, new_path, steps=1000, minibatch_size=512, learning_rate=0.001, blocks=12, filters=128,
          reference_bn_affine=False, device=None, log=print):
    random.seed(123456789)                                                # train.py:103
    entries = load_entries(games_paths)
    ply_count = sum(len(e["moves"]) for e in entries)
    log("Found %i games with %i plies." % (len(entries), ply_count))
    test_entries, train_entries = entries[:10], entries[10:]
    device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if old_path is not None:
        log("Loading old model.")
        cw, bnp = model.load_model(old_path)
        blocks, filters = (len(cw) - 5) // 2, int(cw[0].shape[-1])
    else:
        log("WARNING: Not loading a previous model!")
        cw, bnp = model.random_init(blocks, filters, seed=random.randrange(1 << 30))
    net = Network(blocks, filters, reference_bn_affine).to(device)
    net.load_numpy(cw, bnp)
    opt = torch.optim.SGD(net.parameters(), lr=learning_rate, momentum=0.9)   # tf.train.MomentumOptimizer (model.py:99-100)

    def to_dev(batch):
        return tuple(torch.from_numpy(a).to(device) for a in batch)

    random.seed(123456789)                                                # train.py:131
    val_set = to_dev(make_minibatch(test_entries, 2048))
    log("")
    log("Model dimensions: %i filters, %i blocks, %i parameters." % (
        filters, blocks, sum(int(np.prod(a.shape)) for a in cw)))
    log("Have %i augmented samples, and sampling %i in total." % (ply_count * 8, steps * minibatch_size))
    log("=== BEGINNING TRAINING ===")
    for step_number in range(steps):
        if step_number % 100 == 0:
            net.eval()
            with torch.no_grad():
                pl, vl, _ = losses(net, *val_set)
            log("Step: %4i -- loss: %.6f  (policy: %.6f  value: %.6f)" % (step_number, float(pl + vl), float(pl), float(vl)))
        net.train()
        batch = to_dev(make_minibatch(train_entries, minibatch_size))
        pl, vl, reg = losses(net, *batch)
        opt.zero_grad(set_to_none=True)
        (pl + vl + reg).backward()
        opt.step()
    model.save_model(new_path, *net.to_numpy())
    log("\x1b[35mSaved model to:\x1b[0m %s" % new_path)
    return net
