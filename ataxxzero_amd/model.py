"""The reference's `.npy` weight file layout (model.py:179-196) and random-init nets.

File = np.save(path, [x_conv_weights, x_bn_params]) (model.py:182): a pickled object
array of two lists.
  [0] 2B+5 float32 arrays in creation order (model.py:58-62,67,73,76-77):
      (3,3,4,F), 2B x (3,3,F,F), policy (1,1,F,17), value conv (1,1,F,1), fc_w (49,1), fc_b (1,)
  [1] 2*(2B+1) arrays (F,): moving_mean, moving_variance per batch-norm (model.py:173-177)
Batch-norm gamma/beta are NOT in the file, so every loaded net runs with gamma = 1,
beta = 0.  B and F are inferred from the file.
"""
import numpy as np

BOARD_SIZE = 7
MOVE_TYPES = 17
INPUT_FEATURE_COUNT = 4
BN_EPSILON = 1e-3  # tf.layers.batch_normalization default (no epsilon passed at model.py:120-124)


def layer_shapes(blocks, filters):
    shapes = [(3, 3, INPUT_FEATURE_COUNT, filters)]
    shapes += [(3, 3, filters, filters)] * (2 * blocks)
    shapes += [(1, 1, filters, MOVE_TYPES), (1, 1, filters, 1), (BOARD_SIZE * BOARD_SIZE, 1), (1,)]
    return shapes


def load_model(path):
    """-> (conv_weights, bn_params), validated against model.py's layout."""
    data = np.load(path, allow_pickle=True)
    conv, bn = list(data[0]), list(data[1])
    if len(conv) < 5 or (len(conv) - 5) % 2:
        raise ValueError("%s: %d parameter arrays is not 2B+5" % (path, len(conv)))
    blocks = (len(conv) - 5) // 2
    filters = int(np.asarray(conv[0]).shape[-1])
    for a, shape in zip(conv, layer_shapes(blocks, filters)):
        if tuple(np.asarray(a).shape) != shape:
            raise ValueError("%s: parameter shape %r, expected %r" % (path, np.asarray(a).shape, shape))
    if len(bn) != 2 * (2 * blocks + 1):
        raise ValueError("%s: bad batch normalization parameter count %d" % (path, len(bn)))
    conv = [np.asarray(a, dtype=np.float32) for a in conv]
    bn = [np.asarray(a, dtype=np.float32).reshape(filters) for a in bn]
    return conv, bn


def save_model(path, conv_weights, bn_params):
    # numpy >= 1.24 refuses the reference's ragged np.save([...]); build the object array.
    arr = np.empty(2, dtype=object)
    arr[0] = [np.asarray(a, dtype=np.float32) for a in conv_weights]
    arr[1] = [np.asarray(a, dtype=np.float32) for a in bn_params]
    with open(path, "wb") as f:  # np.save would append ".npy" to a bare path
        np.save(f, arr, allow_pickle=True)


def _truncated_normal(rng, shape, stddev):
    # tf.truncated_normal: redraw samples further than 2 stddev from the mean
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def random_init(blocks=12, filters=128, seed=1, perturb_bn=False):
    """Random-init net with model.py:103-114's distributions: weights truncated normal with
    stddev 0.2*sqrt(2/fan_in), bias 0.01, moving_mean 0, moving_variance 1.  `perturb_bn`
    draws non-trivial statistics so parity tests exercise the batch-norm arithmetic."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = layer_shapes(blocks, filters)
    conv = []
    for shape in shapes[:-1]:
        fan_in = int(np.prod(shape[:-1]))
        conv.append(_truncated_normal(rng, shape, 0.2 * (2.0 / fan_in) ** 0.5))
    conv.append(np.full((1,), 0.01, dtype=np.float32))
    bn = []
    for _ in range(2 * blocks + 1):
        if perturb_bn:
            bn.append((0.1 * rng.standard_normal(filters)).astype(np.float32))
            bn.append(rng.uniform(0.5, 1.5, filters).astype(np.float32))
        else:
            bn.append(np.zeros(filters, dtype=np.float32))
            bn.append(np.ones(filters, dtype=np.float32))
    return conv, bn


def flops_per_eval(blocks, filters):
    """Padded-tap convention of SURVEY.md §8 a7: 2*49*(36F + 18B*F^2 + 18F) + 98."""
    return 2 * 49 * (36 * filters + 18 * blocks * filters * filters + 18 * filters) + 98
