"""Network parity: the fused HIP tower against the float64 numpy restatement of
model.py (oracle/net_oracle.py) on identical boards.

Tolerances: f32 MFMA path <= 1e-5 absolute on logits and value (BASELINE.json
north_star); bf16 / f16 paths are reduced-precision by construction and are bounded
against the f32 path (2e-3 / 3e-4 absolute on logits of scale 0.1: three times the measured error), reported, not
claimed."""
import numpy as np
import pytest

from ataxxzero_amd import link, model
from oracle import net_oracle
from oracle import oracle_lib as orc
from tests.helpers import BLOCK4_MASK

pytestmark = pytest.mark.gpu


def sample_leaf_boards(n, seed, blockers):
    p = orc.pos_from_fen(orc.START_FEN_PLAIN)
    plies, results, boards, moves = link.random_play(64, seed, p.pieces[0], p.pieces[1], blockers, 0, 300)
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        g = int(rng.integers(0, 64))
        ply = int(rng.integers(0, max(plies[g], 1)))
        x, o = int(boards[g, ply, 0]), int(boards[g, ply, 1])
        out.append([x, o] if ply % 2 == 0 else [o, x])
    return np.array(out, dtype=np.uint64)


@pytest.mark.parametrize("blocks,n,perturb", [(2, 37, True), (12, 20, False), (8, 7, True)])
def test_f32_tower_within_1e5_of_float64_restatement(blocks, n, perturb):
    conv, bn = model.random_init(blocks, 128, seed=10 + blocks, perturb_bn=perturb)
    lb = sample_leaf_boards(n, 3, BLOCK4_MASK)
    net = link.Net(conv, bn)
    logits, values = net.forward(lb, BLOCK4_MASK, link.DTYPE_F32)
    ref_p, ref_v = net_oracle.forward(conv, bn, net_oracle.features_from_leaf_boards(lb, BLOCK4_MASK))
    assert np.abs(logits - ref_p).max() <= 1e-5, np.abs(logits - ref_p).max()
    assert np.abs(values - ref_v).max() <= 1e-5
    assert np.abs(ref_p).max() > 1e-3  # the comparison is not vacuous


def game_positions(n, seed, blockers, games=2000):
    """`n` (mover, opponent) boards drawn with numpy PCG64(seed) from `games` uniformly random games on the GPU — config 1's
    game file ("4096 positions drawn (seed 4) from the C1 game file, both sides to move")."""
    p = orc.pos_from_fen(orc.START_FEN_PLAIN)
    plies, results, boards, moves = link.random_play(games, seed, p.pieces[0], p.pieces[1], blockers, 0, 400)
    rng = np.random.default_rng(seed)
    g = rng.integers(0, games, size=n)
    ply = (rng.random(n) * np.maximum(plies[g], 1)).astype(np.int64)
    xo = boards[g, ply]
    odd = (ply % 2 == 1)
    assert 0.3 < odd.mean() < 0.7                           # both sides to move
    return np.where(odd[:, None], xo[:, ::-1], xo).astype(np.uint64)


def note(text):
    """numbers a green run should leave behind (pytest -q swallows prints): appended to gpurun_out/nn_gate.txt"""
    import os
    print(text)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/nn_gate.txt", "a") as f:
        f.write(text + "\n")


def oracle_forward_chunked(conv, bn, lb, blockers, chunk=512):
    ps, vs = [], []
    for lo in range(0, len(lb), chunk):
        p, v = net_oracle.forward(conv, bn, net_oracle.features_from_leaf_boards(lb[lo:lo + chunk], blockers))
        ps.append(p)
        vs.append(v)
    return np.concatenate(ps), np.concatenate(vs)


@pytest.mark.parametrize("blocks,net_seed,blockers", [(12, 1, 0), (8, 3, BLOCK4_MASK)])
def test_f32_tower_within_1e5_on_4096_game_positions(blocks, net_seed, blockers):
    """The north_star gate at the size SURVEY 8(d) gives it: 4096 positions from real (random-play) games, both sides to
    move, the bench's own nets (seed 1: 12x128, seed 3: 8x128, model.py:103-114 initialisation)."""
    conv, bn = model.random_init(blocks, 128, seed=net_seed)
    lb = game_positions(4096, 4, blockers)
    logits, values = link.Net(conv, bn).forward(lb, blockers, link.DTYPE_F32)
    ref_p, ref_v = oracle_forward_chunked(conv, bn, lb, blockers)
    err_p, err_v = np.abs(logits - ref_p).max(), np.abs(values - ref_v).max()
    note("f32 tower vs float64 restatement, %dx128 seed %d, 4096 positions: max |dlogit| %.3e, max |dvalue| %.3e "
         "(max |logit| %.3e)" % (blocks, net_seed, err_p, err_v, np.abs(ref_p).max()))
    assert err_p <= 1e-5 and err_v <= 1e-5
    assert np.abs(ref_p).max() > 1e-3 and len(np.unique(lb, axis=0)) > 3000


def test_f32_tower_on_a_trained_net_at_4096_positions(tmp_path):
    """Random-init nets have logits of scale 0.1 (0.2 x He initialisation); a trained net's are one to two orders of
    magnitude wider, and its batch-norm statistics are no longer (0, 1).  The 12x128 net is trained here the way looper.py
    trains it — self-play games of the random net on the GPU, then `train.py`'s loop — and the gate is run on the result:
    1e-5 x max(1, max |logit|), absolute error reported; the 16-bit towers' errors beside it."""
    import json
    from ataxxzero_amd import selfplay
    conv, bn = model.random_init(12, 128, seed=1)
    sp = selfplay.SelfPlay(conv, bn, games=1024, visits=50, dtype="bf16", seed=5, flags=link.FLAG_EVAL_CACHE, select_budget=64)
    lines = []
    for _ in range(400):
        sp.run(100)
        lines += sp.drain()
        if len(lines) >= 600:
            break
    sp.close()
    assert len(lines) >= 600
    games = str(tmp_path / "games.json")
    with open(games, "w") as f:
        f.write(b"\n".join(lines).decode() + "\n")
    old, new = str(tmp_path / "model-001.npy"), str(tmp_path / "model-002.npy")
    model.save_model(old, conv, bn)
    # (train.py in a process of its own, as looper.py runs it: torch's HIP runtime and this library's do not share a process)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--steps", "400", "--games", games, "--old-path", old,
                          "--new-path", new], cwd=root, capture_output=True, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    conv2, bn2 = model.load_model(new)
    assert max(np.abs(np.asarray(b) - (i % 2)).max() for i, b in enumerate(bn2)) > 0.05   # the statistics have moved
    # half of the boards from the C1 games, half from the self-play games the net was trained on (4 blockers on the board)
    entries = [json.loads(l) for l in lines[:300]]
    own = []
    rng = np.random.default_rng(4)
    while len(own) < 2048:
        e = entries[int(rng.integers(0, len(entries)))]
        ply = int(rng.integers(0, len(e["boards"])))
        cells = e["boards"][ply]
        x = sum(1 << (c % 7 + 7 * (6 - c // 7)) for c in range(49) if cells[c] == 1)
        o = sum(1 << (c % 7 + 7 * (6 - c // 7)) for c in range(49) if cells[c] == 2)
        own.append([x, o] if ply % 2 == 0 else [o, x])
    lb = np.concatenate([game_positions(2048, 4, BLOCK4_MASK), np.array(own, dtype=np.uint64)])
    net = link.Net(conv2, bn2)
    logits, values = net.forward(lb, BLOCK4_MASK, link.DTYPE_F32)
    ref_p, ref_v = oracle_forward_chunked(conv2, bn2, lb, BLOCK4_MASK)
    scale = float(np.abs(ref_p).max())
    err_p, err_v = np.abs(logits - ref_p).max(), np.abs(values - ref_v).max()
    report = ["trained 12x128 net (400 steps on %d self-play games), 4096 positions: max |logit| %.3f, max |value| %.3f"
              % (len(lines), scale, np.abs(ref_v).max()),
              "  f32 tower vs float64 restatement: max |dlogit| %.3e (%.2e of the scale), max |dvalue| %.3e"
              % (err_p, err_p / scale, err_v)]
    note("\n".join(report))
    report = []
    assert scale > 0.5                                         # not a near-zero net
    assert err_p <= 1e-5 * max(1.0, scale) and err_v <= 1e-5
    for name, dt in (("bf16", link.DTYPE_BF16), ("f16", link.DTYPE_F16)):
        p, v = net.forward(lb, BLOCK4_MASK, dt)
        top = (p.reshape(len(p), -1).argmax(1) == ref_p.reshape(len(p), -1).argmax(1)).mean()
        report.append("  %s tower: max |dlogit| %.3e, max |dvalue| %.3e, same arg-max logit on %.1f %% of the boards"
                      % (name, np.abs(p - ref_p).max(), np.abs(v - ref_v).max(), 100 * top))
        assert np.abs(p - ref_p).max() <= (4e-2 if dt == link.DTYPE_BF16 else 5e-3) * max(1.0, scale)
    note("\n".join(report))


@pytest.mark.parametrize("filters,blocks,n", [(64, 3, 29), (256, 2, 10), (256, 4, 7)])
def test_other_filter_counts(filters, blocks, n):
    """model.Network.FILTERS is a class attribute the reference patches (uai_interface.py:92-93): 64- and 256-filter nets
    run on the width-templated 32x32 tower — f32 within 1e-5 of the float64 restatement, 16-bit paths tracking it, batch
    position irrelevant, and the search engine plays with them."""
    conv, bn = model.random_init(blocks, filters, seed=filters + blocks, perturb_bn=True)
    assert conv[0].shape == (3, 3, 4, filters)
    lb = sample_leaf_boards(n, 4, BLOCK4_MASK)
    net = link.Net(conv, bn)
    assert net.filters == filters
    logits, values = net.forward(lb, BLOCK4_MASK, link.DTYPE_F32)
    ref_p, ref_v = net_oracle.forward(conv, bn, net_oracle.features_from_leaf_boards(lb, BLOCK4_MASK))
    assert np.abs(logits - ref_p).max() <= 1e-5, np.abs(logits - ref_p).max()
    assert np.abs(values - ref_v).max() <= 1e-5 and np.abs(ref_p).max() > 1e-3
    scale = float(np.abs(ref_p).max())
    for dtype, rel in ((link.DTYPE_BF16, 4e-2), (link.DTYPE_F16, 5e-3)):
        p, v = net.forward(lb, BLOCK4_MASK, dtype)
        assert np.abs(p - logits).max() <= rel * scale and np.abs(v - values).max() <= rel
    part_p, part_v = net.forward(lb[:4], BLOCK4_MASK, link.DTYPE_F32)
    assert (part_p == logits[:4]).all() and (part_v == values[:4]).all()
    sp, sv = net.forward_sym(lb[:3], BLOCK4_MASK, link.DTYPE_F32)
    rp, rv = net_oracle.forward_sym(conv, bn, net_oracle.features_from_leaf_boards(lb[:3], BLOCK4_MASK))
    assert np.abs(sp - rp).max() <= 1e-5 and np.abs(sv - rv).max() <= 1e-5
    ocfg = orc.make_config(40, 10, seed=2)
    ge = link.Engine(link.Config(**{k: getattr(ocfg, k) for k, _ in orc.Config._fields_}))
    ge.run(net, 120, link.DTYPE_BF16)
    ge.sync()
    st = ge.stats()
    assert st["plies"] > 40 * 5 and st["edge_overflow"] == 0
    # ... and the device-resident loop with this width's f32 tower (whose launch also carries the loop's move-playing
    # workgroups) stays the oracle's search bit for bit
    from tests.test_gpu_engine import _oracle_follow, compare_all
    oe, ge = orc.Engine(ocfg), link.Engine(link.Config(**{k: getattr(ocfg, k) for k, _ in orc.Config._fields_}))
    ge.run(net, 150, link.DTYPE_F32)
    _oracle_follow(oe, net, ocfg.blockers, 150)
    ge.sync()
    compare_all(oe, ge, range(40))
    assert oe.stats()["plies"] > 40 * 5
    with pytest.raises(link.AzhError):
        link.Net(*model.random_init(1, 96, seed=1))   # only 64 / 128 / 256 are built


def test_zero_slot_build_of_the_tower_equals_the_default_bit_for_bit(tmp_path):
    """The default 16-bit tower reads off-board taps as out-of-range LDS addresses (AZH_OOBZERO=1: the hardware's range
    check supplies the zeros; probed per device).  Its escape hatch — the -DAZH_OOBZERO=0 build, off-board taps steered to
    zero slots inside the allocation — is compiled here and must give the same bits on the same boards, bf16 and f16:
    the two differ only in WHERE a zero is read."""
    import os
    import shutil
    import subprocess
    import sys
    from ataxxzero_amd import build
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    lib = build.build_variant("oobzero0", ["-DAZH_OOBZERO=0"])
    assert os.path.exists(lib) and os.path.abspath(lib) != os.path.abspath(link.library_path())
    conv, bn = model.random_init(3, 128, seed=41, perturb_bn=True)
    lb = sample_leaf_boards(200, 12, BLOCK4_MASK)   # 66 full workgroups and a partial one
    weights, boards, out = str(tmp_path / "w.npy"), str(tmp_path / "boards.npy"), str(tmp_path / "out.npz")
    model.save_model(weights, conv, bn)
    np.save(boards, lb)
    script = ("import sys, numpy as np\n"
              "sys.path.insert(0, %r)\n"
              "from ataxxzero_amd import link, model\n"
              "assert link.library_path() != %r and link.load()._name == %r\n"
              "net = link.Net(*model.load_model(%r))\n"
              "lb = np.load(%r)\n"
              "res = {}\n"
              "for name, dt in (('bf16', link.DTYPE_BF16), ('f16', link.DTYPE_F16)):\n"
              "    p, v = net.forward(lb, %d, dt)\n"
              "    res[name + '_p'], res[name + '_v'] = p, v\n"
              "np.savez(%r, **res)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), lib, lib, weights,
                                          boards, BLOCK4_MASK, out)
    env = dict(os.environ, AZH_LIB=lib)
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, timeout=600)
    assert res.returncode == 0, res.stdout.decode()[-2000:] + res.stderr.decode()[-2000:]
    other = np.load(out)
    net = link.Net(conv, bn)
    for name, dt in (("bf16", link.DTYPE_BF16), ("f16", link.DTYPE_F16)):
        p, v = net.forward(lb, BLOCK4_MASK, dt)
        assert np.abs(p).max() > 1e-3
        assert (p.view(np.uint32) == other[name + "_p"].view(np.uint32)).all(), name
        assert (v.view(np.uint32) == other[name + "_v"].view(np.uint32)).all(), name


def test_first_boards_of_partial_tiles_and_empty_batch():
    conv, bn = model.random_init(1, 128, seed=4, perturb_bn=True)
    net = link.Net(conv, bn)
    lb = sample_leaf_boards(13, 8, 0)
    full_p, full_v = net.forward(lb, 0, link.DTYPE_F32)
    for k in (1, 2, 3, 4, 7):
        p, v = net.forward(lb[:k], 0, link.DTYPE_F32)
        assert (p == full_p[:k]).all() and (v == full_v[:k]).all()  # batch position does not change a board's result
    p, v = net.forward(lb[:0], 0, link.DTYPE_F32)
    assert p.shape == (0, 7, 7, 17)


# tolerances = 3x the error measured on this net (tools/precision_in_the_loop.py, profiles/round2_precision_in_the_loop.json:
# max |dlogit| 6.9e-4 bf16 / 9.6e-5 f16 on logits of scale 0.1; what that does to the search is reported there)
@pytest.mark.parametrize("dtype,tol", [(link.DTYPE_BF16, 2e-3), (link.DTYPE_F16, 3e-4)])
def test_reduced_precision_tower_tracks_f32(dtype, tol):
    conv, bn = model.random_init(12, 128, seed=1)
    lb = sample_leaf_boards(50, 5, BLOCK4_MASK)
    net = link.Net(conv, bn)
    p32, v32 = net.forward(lb, BLOCK4_MASK, link.DTYPE_F32)
    p, v = net.forward(lb, BLOCK4_MASK, dtype)
    err_p, err_v = np.abs(p - p32).max(), np.abs(v - v32).max()
    print("dtype %d: max |dlogit| %.3e, max |dvalue| %.3e, logit scale %.3e" % (dtype, err_p, err_v, np.abs(p32).max()))
    assert err_p <= tol and err_v <= tol
    # same argmax move on nearly every board
    same = (p.reshape(len(p), -1).argmax(1) == p32.reshape(len(p), -1).argmax(1)).mean()
    assert same >= 0.8


def test_symmetry_averaged_forward_matches_nn_evals_restatement():
    """azh_net_forward_sym == nn_evals.evaluate (nn_evals.py:48-62) as restated in float64; an asymmetric
    blocker mask checks that the blocker plane is transformed with the board."""
    conv, bn = model.random_init(2, 128, seed=21, perturb_bn=True)
    net = link.Net(conv, bn)
    for blockers in (BLOCK4_MASK, (1 << 0) | (1 << 9) | (1 << 33)):
        lb = sample_leaf_boards(11, 6, blockers)
        p, v = net.forward_sym(lb, blockers, link.DTYPE_F32)
        feats = net_oracle.features_from_leaf_boards(lb, blockers)
        ref_p, ref_v = net_oracle.forward_sym(conv, bn, feats)
        assert np.abs(p - ref_p).max() <= 1e-5, np.abs(p - ref_p).max()
        assert np.abs(v - ref_v).max() <= 1e-5
        plain_p, _ = net_oracle.forward(conv, bn, feats)
        assert np.abs(plain_p - ref_p).max() > 1e-3  # averaging changes the answer: the test is not vacuous
        p16, v16 = net.forward_sym(lb, blockers, link.DTYPE_BF16)
        assert np.abs(p16 - p).max() < 5e-2 and np.abs(v16 - v).max() < 5e-2
    p0, v0 = net.forward_sym(lb[:0], 0, link.DTYPE_F32)
    assert p0.shape == (0, 7, 7, 17)


@pytest.mark.parametrize("dtype,tol", [(link.DTYPE_BF16, 2e-3), (link.DTYPE_F16, 3e-4)])
def test_thin_tower_equals_the_three_board_tower_to_rounding_and_is_position_independent(dtype, tol):
    """One board per workgroup (azh_net_forward_thin: the kernel for a match's last games and a UAI engine's single
    position) computes model.Network's forward (model.py:38-79) for the same boards: equal to the 3-board tower within the
    16-bit towers' own rounding (another summation order), bounded against the f32 tower like it, and bit-identical
    wherever a board sits in the launch and whatever is beside it — empty batch, one board, a ragged count."""
    conv, bn = model.random_init(12, 128, seed=1, perturb_bn=True)
    net = link.Net(conv, bn)
    boards = sample_leaf_boards(301, 5, BLOCK4_MASK)
    p3, v3 = net.forward(boards, BLOCK4_MASK, dtype)
    p1, v1 = net.forward(boards, BLOCK4_MASK, dtype, thin=True)
    pf, vf = net.forward(boards, BLOCK4_MASK, link.DTYPE_F32)
    assert np.abs(p1 - p3).max() <= tol and np.abs(v1 - v3).max() <= tol
    assert np.abs(p1 - pf).max() <= np.abs(p3 - pf).max() * 1.5 + 1e-6 and np.abs(v1 - vf).max() <= np.abs(v3 - vf).max() * 1.5 + 1e-6
    # any order, any neighbours: the same bits
    perm = np.random.default_rng(3).permutation(len(boards))
    pq, vq = net.forward(boards[perm], BLOCK4_MASK, dtype, thin=True)
    assert (pq == p1[perm]).all() and (vq == v1[perm]).all()
    for lo, n in ((0, 1), (7, 2), (100, 37)):
        ps, vs = net.forward(boards[lo:lo + n], BLOCK4_MASK, dtype, thin=True)
        assert (ps == p1[lo:lo + n]).all() and (vs == v1[lo:lo + n]).all()
    pe, ve = net.forward(boards[:0], BLOCK4_MASK, dtype, thin=True)
    assert pe.shape[0] == 0 and ve.shape[0] == 0
    # f32 has no thin kernel: the call runs the ordinary f32 tower
    pt, vt = net.forward(boards[:9], BLOCK4_MASK, link.DTYPE_F32, thin=True)
    assert (pt == pf[:9]).all() and (vt == vf[:9]).all()


@pytest.mark.parametrize("dtype", [link.DTYPE_BF16, link.DTYPE_F16, link.DTYPE_F32])
def test_full_size_launch_is_position_independent(dtype):
    """A 16384-board launch (the bench's largest): a board's logits and value do not depend on which workgroup
    computes it or on its neighbours — bit for bit at equal board slot (index mod 3).  The f32 tower is bit-identical in
    every slot; the 16-bit tower sums the taps of the two wave halves in different orders (compile-time tap skipping),
    so across slots it agrees to rounding only."""
    conv, bn = model.random_init(12, 128, seed=1)
    net = link.Net(conv, bn)
    base = sample_leaf_boards(1021, 9, BLOCK4_MASK)          # 1021 is prime: every board meets every slot of a workgroup
    big = np.concatenate([base] * 17)[:16384]
    p, v = net.forward(big, BLOCK4_MASK, dtype)
    assert np.isfinite(p).all() and np.isfinite(v).all() and (np.abs(v) <= 1).all()
    tol = {link.DTYPE_BF16: 2e-3, link.DTYPE_F16: 3e-4, link.DTYPE_F32: 0.0}[dtype]
    for rep in range(1, 16):
        lo = rep * 1021
        n = min(1021, 16384 - lo)
        if lo % 3 == 0 or dtype == link.DTYPE_F32:
            assert (p[lo:lo + n] == p[:n]).all() and (v[lo:lo + n] == v[:n]).all()
        else:
            assert np.abs(p[lo:lo + n] - p[:n]).max() <= tol and np.abs(v[lo:lo + n] - v[:n]).max() <= tol
    small_p, small_v = net.forward(base[:37], BLOCK4_MASK, dtype)
    assert (small_p == p[:37]).all() and (small_v == v[:37]).all()


def test_f32_tower_agrees_with_pytorch_conv_and_batch_norm():
    """A third-party arithmetic beside the numpy restatement: the same weights in torch.nn (Conv2d / BatchNorm2d in
    eval mode / Linear, float64 on the CPU, NCHW) and the HIP f32 tower must agree to 1e-5 on identical boards.  The
    reference's own arithmetic (TensorFlow 1) is not available; two independent implementations of its documented
    semantics are the closest check there is."""
    import torch
    from ataxxzero_amd import training
    conv, bn = model.random_init(12, 128, seed=3, perturb_bn=True)
    tnet = training.Network(12, 128).double()
    tnet.load_numpy(conv, bn)
    tnet.eval()
    lb = sample_leaf_boards(24, 11, BLOCK4_MASK)
    feats = net_oracle.features_from_leaf_boards(lb, BLOCK4_MASK, np.float64)
    with torch.no_grad():
        tp, tv = tnet(torch.from_numpy(feats).permute(0, 3, 1, 2))
    p, v = link.Net(conv, bn).forward(lb, BLOCK4_MASK, link.DTYPE_F32)
    assert np.abs(p - tp.numpy()).max() <= 1e-5 and np.abs(v - tv.numpy()).max() <= 1e-5
    assert np.abs(tp.numpy()).max() > 1e-3
