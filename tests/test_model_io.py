"""`.npy` weight layout (model.py:179-196) and random-init distributions (model.py:103-114)."""
import numpy as np
import pytest

from ataxxzero_amd import model
from oracle import net_oracle


def test_roundtrip_in_reference_layout(tmp_path):
    conv, bn = model.random_init(3, 128, seed=5, perturb_bn=True)
    path = str(tmp_path / "model-001.npy")
    model.save_model(path, conv, bn)
    raw = np.load(path, allow_pickle=True)  # what model.load_model does (model.py:187)
    assert raw.shape == (2,) and len(raw[0]) == 2 * 3 + 5 and len(raw[1]) == 2 * (2 * 3 + 1)
    assert [a.shape for a in raw[0]] == model.layer_shapes(3, 128)
    conv2, bn2 = model.load_model(path)
    assert all((a == b).all() for a, b in zip(conv, conv2)) and all((a == b).all() for a, b in zip(bn, bn2))


def test_rejects_malformed_files(tmp_path):
    conv, bn = model.random_init(1, 128, seed=1)
    path = str(tmp_path / "bad.npy")
    model.save_model(path, conv[:-1], bn)
    with pytest.raises(ValueError):
        model.load_model(path)
    model.save_model(path, conv, bn[:-1])
    with pytest.raises(ValueError):
        model.load_model(path)


def test_random_init_distributions():
    conv, bn = model.random_init(12, 128, seed=1)
    assert sum(a.size for a in conv) == 3545906  # SURVEY.md §8 a7 parameter count
    w = conv[3]
    sd = 0.2 * (2.0 / (3 * 3 * 128)) ** 0.5
    assert np.abs(w).max() <= 2 * sd + 1e-9 and abs(w.std() / sd - 0.88) < 0.02  # truncated at 2 sigma
    assert conv[-1].tolist() == [pytest.approx(0.01)]
    assert all((m == 0).all() for m in bn[0::2]) and all((v == 1).all() for v in bn[1::2])
    assert model.flops_per_eval(12, 128) == 347493986 and model.flops_per_eval(8, 128) == 231888482


def test_net_oracle_shapes_and_value_range():
    conv, bn = model.random_init(2, 128, seed=2, perturb_bn=True)
    lb = np.array([[(1 << 42) | (1 << 6), (1 << 48) | 1], [(1 << 48) | 1, (1 << 42) | (1 << 6)]], dtype=np.uint64)
    f = net_oracle.features_from_leaf_boards(lb, 0)
    assert f.shape == (2, 7, 7, 4) and f[0, 0, 0].tolist() == [1, 1, 0, 0] and f[1, 0, 0].tolist() == [1, 0, 1, 0]
    p, v = net_oracle.forward(conv, bn, f)
    assert p.shape == (2, 7, 7, 17) and v.shape == (2, 1) and (np.abs(v) < 1).all()
    p32, v32 = net_oracle.forward(conv, bn, f, dtype=np.float32)
    assert np.abs(p - p32).max() < 1e-4
