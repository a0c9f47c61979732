"""train.py port: the torch network equals the numpy restatement of model.py, the sample
pipeline follows train.py:43-77, and a short run writes a loadable .npy (CPU torch here)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from ataxxzero_amd import model, training
from oracle import net_oracle
from oracle import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_games(n, visits=6, seed=2):
    """Self-play style entries (with dists) from the oracle engine with a null evaluator."""
    e = orc.Engine(orc.make_config(games=8, visits=visits, seed=seed, fen_str=orc.START_FEN_PLAIN))
    games = []
    z, v = np.zeros((8, 833), np.float32), np.zeros(8, np.float32)
    while len(games) < n:
        e.select()
        e.backup(z, v)
        games += [g["entry"] for g in e.pop_games()]
    return games[:n]


def test_torch_network_matches_numpy_restatement():
    conv, bn = model.random_init(2, 128, seed=7, perturb_bn=True)
    net = training.Network(2, 128)
    net.load_numpy(conv, bn)
    net.eval()
    rng = np.random.default_rng(0)
    lb = np.array([[(1 << 42) | (1 << 6) | (1 << 20), (1 << 48) | 1 | (1 << 30)]] * 3, dtype=np.uint64)
    lb[1, 0] |= np.uint64(1 << 10)
    lb[2] = lb[2, ::-1]
    feats = net_oracle.features_from_leaf_boards(lb, 0, np.float32)
    with torch.no_grad():
        p, v = net(torch.from_numpy(feats).permute(0, 3, 1, 2))
    ref_p, ref_v = net_oracle.forward(conv, bn, feats)
    assert np.abs(p.numpy() - ref_p).max() < 1e-5 and np.abs(v.numpy() - ref_v).max() < 1e-5
    cw, bnp = net.to_numpy()
    assert all(np.array_equal(a, b) for a, b in zip(cw, conv)) and all(np.array_equal(a, b) for a, b in zip(bnp, bn))


def test_sample_pipeline_targets():
    games = oracle_games(3)
    random.seed(5)
    for _ in range(50):
        f, p, v = training.get_sample_from_entries(games)
        assert f.shape == (7, 7, 4) and (f[:, :, 0] == 1).all() and f[:, :, 3].sum() == 0
        assert abs(p.sum() - 1) < 1e-3 and p.min() >= 0 and v[0] in (1, -1)
    # symmetries: a clone to (1, 0) under coin1 (x flip) lands on (5, 0); coin3 swaps axes
    assert training.apply_symmetry_to_move(1, ("c", (1, 0))) == ("c", (5, 0))
    assert training.apply_symmetry_to_move(4, ((0, 2), (2, 3))) == ((2, 0), (3, 2))
    arr = np.arange(7 * 7 * 2).reshape(7, 7, 2)
    assert (training.apply_symmetry(1, arr) == arr[::-1]).all() and (training.apply_symmetry(4, arr) == arr.swapaxes(0, 1)).all()
    # the heat-map index convention equals the engine's policy index (oracle): jump a7c6 -> layer of (+2,+1)
    hm = np.zeros((7, 7, 17), np.float32)
    training.add_move_to_heatmap(hm, training.uai_decode_move("a7c6"))
    assert int(np.flatnonzero(hm.ravel())[0]) == orc.lib().orc_policy_index(orc.move_from_string("a7c6")) == 269
    # python-generator entries (no dists, nested-list moves) give one-hot targets
    e = {"boards": [[0] * 49], "moves": [["c", [3, 3]]], "result": 1}
    e["boards"][0][0] = 1
    random.seed(1)
    f, p, v = training.get_sample_from_entries([e])
    assert p.sum() == 1 and p.max() == 1 and v == [1]


def test_train_cli_writes_reference_layout(tmp_path):
    games = oracle_games(14)
    gpath = str(tmp_path / "model-001-0.json")
    with open(gpath, "w") as f:
        for g in games:
            f.write(json.dumps(g) + "\n")
    conv, bn = model.random_init(1, 128, seed=3)
    old, new = str(tmp_path / "model-001.npy"), str(tmp_path / "model-002.npy")
    model.save_model(old, conv, bn)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--steps", "3", "--minibatch-size", "16", "--games", gpath,
                          "--old-path", old, "--new-path", new], cwd=ROOT, capture_output=True, timeout=600,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    out = res.stdout.decode()
    assert "Found 14 games" in out and "Step:    0 -- loss:" in out and "=== BEGINNING TRAINING ===" in out
    conv2, bn2 = model.load_model(new)
    assert [a.shape for a in conv2] == [a.shape for a in conv]
    assert any(not np.array_equal(a, b) for a, b in zip(conv, conv2))       # weights moved
    assert not np.array_equal(bn2[0], bn[0]) and (np.asarray(bn2[1]) > 0).all()  # moving statistics updated


def test_batched_minibatch_is_bit_identical_to_the_sample_by_sample_pipeline():
    """training.make_minibatch assembles a minibatch with array operations; it must draw from `random` exactly as the
    reference's per-sample loop does (train.py:123-130) and give the same arrays bit for bit, for all entry flavours
    (dists / one-hot nested-list moves / random_ply)."""
    import json
    import os
    import random
    from tests.helpers import GOLDEN
    with open(os.path.join(GOLDEN, "train_entries.json")) as f:
        entries = json.load(f)
    for seed in range(6):
        random.seed(seed)
        a = training.make_minibatch_reference(entries, 200)
        state_a = random.getstate()
        random.seed(seed)
        b = training.make_minibatch(entries, 200)
        assert random.getstate() == state_a                       # the same number of draws, in the same order
        for x, y in zip(a, b):
            assert x.dtype == y.dtype == np.float32 and x.shape == y.shape and np.array_equal(x, y)
    assert a[0].shape == (200, 7, 7, 4) and a[1].shape == (200, 7, 7, 17) and a[2].shape == (200, 1)
