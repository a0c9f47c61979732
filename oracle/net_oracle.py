"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference network's forward pass (model.py:38-79,116-142)
in numpy.  PARITY UNPINNED at this boundary: the reference arithmetic lives in
TensorFlow 1.x (tf.nn.conv2d, tf.layers.batch_normalization, tf.matmul, tf.nn.tanh;
unpinned version, absent from this image and from /root/reference), and the
reference holds no golden vector or saved model for it.  This file
restates the published semantics of those ops:

  conv2d NHWC/HWIO, stride 1, padding SAME, cross-correlation:
      out[n,x,y,o] = sum_{i,j,c} in[n, x+i-1, y+j-1, c] * W[i,j,c,o]   (zero outside)
  batch_normalization inference: (x - moving_mean) / sqrt(moving_variance + 1e-3),
      gamma = 1, beta = 0 (not saved by model.save_model: SURVEY.md appendix B, Q1)
  value = tanh(reshape(conv1x1(h), [N,49]) @ fc_w + fc_b)

float64 by default (the 1e-5 gate of the HIP f32 path is checked against this).  Imported by tests/ and
__graft_entry__.smoke() only: bench.py's cpu_baseline is tools/cpu_baseline, not this file.
"""
import numpy as np

BN_EPS = 1e-3


def features_from_leaf_boards(leaf_boards, blockers, dtype=np.float64):
    """(n,2) u64 (mover, opponent) -> (n,7,7,4), cpp/self_play_client.cpp:174-202."""
    leaf_boards = np.asarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
    n = len(leaf_boards)
    out = np.zeros((n, 7, 7, 4), dtype=dtype)
    out[..., 0] = 1
    for x in range(7):
        for y in range(7):
            sq = x + 7 * (6 - y)
            out[:, x, y, 1] = (leaf_boards[:, 0] >> np.uint64(sq)) & np.uint64(1)
            out[:, x, y, 2] = (leaf_boards[:, 1] >> np.uint64(sq)) & np.uint64(1)
            out[:, x, y, 3] = (int(blockers) >> sq) & 1
    return out


def conv2d_same(x, w):
    """x (n,7,7,c), w (k,k,c,o) -> (n,7,7,o); model.py:118 / :68 / :74."""
    if w.shape[0] == 1:
        return x @ w[0, 0]
    n, c = x.shape[0], x.shape[3]
    xp = np.zeros((n, 9, 9, c), dtype=x.dtype)
    xp[:, 1:8, 1:8, :] = x
    # im2col with k index (i, j, c), matching w.reshape(9c, o)
    cols = np.concatenate([xp[:, i:i + 7, j:j + 7, :] for i in range(3) for j in range(3)], axis=3)
    return (cols.reshape(n * 49, 9 * c) @ w.reshape(9 * c, w.shape[3])).reshape(n, 7, 7, -1)


def batch_norm(x, mean, var):
    return (x - mean) / np.sqrt(var + BN_EPS)


def forward(conv_weights, bn_params, features, dtype=np.float64):
    """-> (policy logits (n,7,7,17), value (n,1)); model.py:38-79."""
    cw = [np.asarray(a, dtype=dtype) for a in conv_weights]
    bn = [np.asarray(a, dtype=dtype) for a in bn_params]
    blocks = (len(cw) - 5) // 2
    h = np.asarray(features, dtype=dtype)
    h = np.maximum(batch_norm(conv2d_same(h, cw[0]), bn[0], bn[1]), 0)          # model.py:56-57
    for b in range(blocks):                                                      # model.py:132-142
        i1, i2 = 1 + 2 * b, 2 + 2 * b
        t = np.maximum(batch_norm(conv2d_same(h, cw[i1]), bn[2 * i1], bn[2 * i1 + 1]), 0)
        t = batch_norm(conv2d_same(t, cw[i2]), bn[2 * i2], bn[2 * i2 + 1])
        h = np.maximum(t + h, 0)
    policy = conv2d_same(h, cw[2 * blocks + 1])                                  # model.py:66-69
    v = conv2d_same(h, cw[2 * blocks + 2]).reshape(len(h), 49)                   # model.py:71-75
    value = np.tanh(v @ cw[2 * blocks + 3] + cw[2 * blocks + 4])                 # model.py:76-79
    return policy, value


def apply_symmetry(tensor, symmetry):
    """The dihedral image `symmetry` (0..7) of a (7,7,k) tensor as nn_evals.py:8-16 defines it: bit 0 mirrors the first
    axis, bit 1 the second, bit 2 then swaps the two."""
    axes = [axis for axis, bit in ((0, 1), (1, 2)) if symmetry & bit]
    image = np.flip(tensor, axis=axes) if axes else tensor
    return np.swapaxes(image, 0, 1) if symmetry & 4 else image


# inverse of every image under composition (nn_evals.py:27): the two mirror-then-swap images 5 and 6 undo each other
INVERSE_SYMMETRY = {s: (s if s not in (5, 6) else 11 - s) for s in range(8)}


def sym_average(evaluate, features):
    """nn_evals.evaluate (nn_evals.py:48-62) for a batch, around any evaluator `evaluate(images (m,7,7,4)) ->
    (policy (m,7,7,17), value (m,1))`: the evaluator on the 8 dihedral images of every board, policies brought back
    with the inverse symmetry (spatial axes only) and averaged, values averaged.  Pinned by
    tests/golden/nn_evals_sym.npz (the reference's own nn_evals.evaluate around the same injected evaluator)."""
    n = len(features)
    images = np.stack([apply_symmetry(f, s) for f in features for s in range(8)])
    policy, value = evaluate(images)
    policy = np.asarray(policy).reshape(n, 8, 7, 7, 17)
    back = np.stack([np.stack([apply_symmetry(policy[i, s], INVERSE_SYMMETRY[s]) for s in range(8)]) for i in range(n)])
    return back.mean(axis=1), np.asarray(value).reshape(n, 8).mean(axis=1).reshape(n, 1)


def forward_sym(conv_weights, bn_params, features, dtype=np.float64):
    """sym_average around the restated net.  -> (policy (n,7,7,17), value (n,1))."""
    features = np.asarray(features, dtype=dtype)
    return sym_average(lambda images: forward(conv_weights, bn_params, images, dtype=dtype), features)
