"""The HIP engine and rules kernels against vectors produced by the reference's own engine.py
(tests/golden/gen_engine_fixtures.py): feature rows, posterior, and 48 complete PUCT searches whose every edge
(visits exactly, total scores and priors to f32 rounding) must match the tree engine.py built — the arena
semantics pinned to the reference itself rather than to the oracle restatement."""
import os

import numpy as np
import pytest

from ataxxzero_amd import link
from oracle import oracle_lib as orc
from tests import engine_fixture_checks as fx
from tests.helpers import BLOCK4_MASK, GOLDEN, load_gz, synthetic_evals_distinct

pytestmark = pytest.mark.gpu


def gpu_engine(fen, visits):
    ocfg = fx.config_for(fen, visits)
    return link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))


def test_feature_kernel_equals_engine_board_to_features():
    feats = np.load(os.path.join(GOLDEN, "engine_features.npz"))
    for name, blockers in (("noblock", 0), ("block4", BLOCK4_MASK)):
        recs = load_gz("rules_%s.json.gz" % name)
        lb = np.zeros((len(recs), 2), dtype=np.uint64)
        for i, rec in enumerate(recs):
            p = orc.pos_from_fen(rec["fen"])
            lb[i] = (p.pieces[p.turn], p.pieces[1 - p.turn])
        got = link.features_batch(lb, blockers)
        assert got.shape == feats[name].shape and (got == feats[name].astype(np.float32)).all()


def test_posterior_equals_nn_evaluator():
    for rec in fx.posterior_fixtures():
        ge = gpu_engine(rec["fen"], 4)
        assert ge.select() == 1
        need, lb = ge.leaves()
        logits, values = synthetic_evals_distinct(lb)
        assert values[0] == np.float32(rec["value"])
        ge.set_evals(logits, values)
        ge.backup()
        _, root = fx.walk_tree(ge.tree(0))
        fx.check_priors(root, rec["posterior"])
        ge.close()


def test_dirichlet_mix_equals_engine_py():
    """Root priors with Dirichlet noise == engine.add_dirichlet_noise_to_posterior (engine.py:117-124, the mix of
    cpp/self_play_client.cpp:250-271) given the engine's own normalised gamma draws — with the python posterior underneath
    (same expression, f32 rounding) and with the C++ generator's posterior (flags 0)."""
    recs = fx.dirichlet_fixtures()
    assert len(recs) >= 24
    for flags in (orc.FLAG_PY_POSTERIOR, 0):
        for rec in recs:
            ocfg = fx.dirichlet_config(rec, flags)
            ge = link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))
            assert ge.select() == 1 and ge.game_state(0).leaf_kind == link.LEAF_ROOT
            logits, values = synthetic_evals_distinct(ge.leaves()[1])
            ge.set_evals(logits, values)
            ge.backup()
            _, root = fx.walk_tree(ge.tree(0))
            fx.check_dirichlet(rec, root, flags)
            ge.close()


def test_search_reproduces_engine_py_trees():
    recs = fx.mcts_fixtures()
    assert len(recs) == 48
    for rec in recs:
        ge = gpu_engine(rec["fen"], rec["visits"])

        def backup(logits, values):
            ge.set_evals(logits, values)
            ge.backup()

        fx.drive(ge.select, lambda: ge.leaves()[1], backup, 1 + rec["visits"])
        s = ge.game_state(0)
        assert s.ply == 0 and s.phase == 2
        fx.check_search(rec, ge.tree(0), s)
        ge.close()


def test_tree_reuse_across_moves_equals_mcts_play():
    """advance_game's re-root (breadth-first subtree copy into the other arena) against engine.MCTS.play: 16 four-ply
    sequences of forced moves; the tree before every move and the kept subtree after the last one, edge for edge."""
    for rec in fx.reuse_fixtures():
        ocfg = fx.reuse_config(rec)
        ge = link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))

        def backup(logits, values):
            ge.set_evals(logits, values)
            ge.backup()

        fx.check_reuse_sequence(rec, ge.select, lambda: ge.leaves()[1], backup, lambda: ge.game_state(0), lambda: ge.tree(0))
        assert ge.stats()["reroot_nodes"] > len(rec["plies"])
        ge.close()
