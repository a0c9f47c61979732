"""The oracle (and the host-side helpers of the product) against vectors produced by the reference's OWN Python:
engine.py (features, policy indices, posterior, PUCT search, exponent sampling), train.py (sample pipeline) and
nn_evals.py (symmetry averaging), imported unmodified by tests/golden/gen_engine_fixtures.py.  This is what pins
oracle/mcts_oracle.c's Python-engine mode (the arena semantics) and the encoders to the reference
instead of to a restatement."""
import json
import os
import random

import numpy as np

from ataxxzero_amd import training
from oracle import net_oracle
from oracle import oracle_lib as orc
from tests import engine_fixture_checks as fx
from tests.helpers import GOLDEN, linear_evals, load_gz, synthetic_evals_distinct


def test_feature_rows_equal_engine_board_to_features():
    feats = np.load(os.path.join(GOLDEN, "engine_features.npz"))
    total = 0
    for name in ("noblock", "block4"):
        recs = load_gz("rules_%s.json.gz" % name)
        assert len(recs) == len(feats[name])
        for rec, want in zip(recs, feats[name]):
            got = orc.features(orc.pos_from_fen(rec["fen"]))
            assert (got == want.astype(np.float32)).all(), rec["fen"]
            total += 1
    assert total == 3674 and feats["block4"][..., 3].sum() == 4 * len(feats["block4"])


def test_policy_index_of_every_move_equals_get_move_score():
    with open(os.path.join(GOLDEN, "engine_policy_index.json")) as f:
        table = json.load(f)
    assert len(table) == 529  # 49 clones + 480 jumps
    for uai, idx in table.items():
        assert orc.lib().orc_policy_index(orc.move_from_string(uai)) == idx, uai
        heat = np.zeros((7, 7, 17))
        training.add_move_to_heatmap(heat, training.uai_decode_move(uai))
        assert int(np.flatnonzero(heat.ravel())[0]) == idx and heat.sum() == 1
    # every legal move of every fixture position is covered by the table
    for rec in load_gz("rules_noblock.json.gz")[::7]:
        assert all(m in table for m in rec["moves"] if m != "0000")


def _root_eval(fen):
    oe = orc.Engine(fx.config_for(fen, 4))
    n, need = oe.select()
    assert n == 1 and oe.game_state(0).leaf_kind == orc.LEAF_ROOT
    lb = oe.leaf_boards()
    logits, values = synthetic_evals_distinct(lb)
    oe.backup(logits, values)
    return oe, float(values[0])


def test_posterior_equals_nn_evaluator():
    recs = fx.posterior_fixtures()
    assert len(recs) == 96
    for rec in recs:
        oe, value = _root_eval(rec["fen"])
        _, root = fx.walk_tree(oe.tree(0))
        fx.check_priors(root, rec["posterior"])
        assert value == np.float32(rec["value"])
        # (the sum is 1 / (1 + 1e-6 / legal mass), engine.py:202 — not 1)
        assert abs(sum(p for _, p in root) - sum(p for _, p in rec["posterior"])) < 1e-5


def test_dirichlet_mix_equals_engine_py():
    """The root-noise mix of the C++ generator (cpp/self_play_client.cpp:250-271) pinned to engine.py's statement of it
    (engine.py:117-124), fed the engines' own normalised gamma draws."""
    recs = fx.dirichlet_fixtures()
    assert len(recs) >= 24
    for flags in (orc.FLAG_PY_POSTERIOR, 0):
        for rec in recs:
            oe = orc.Engine(fx.dirichlet_config(rec, flags))
            n, need = oe.select()
            assert n == 1 and oe.game_state(0).leaf_kind == orc.LEAF_ROOT
            logits, values = synthetic_evals_distinct(oe.leaf_boards())
            oe.backup(logits, values)
            _, root = fx.walk_tree(oe.tree(0))
            fx.check_dirichlet(rec, root, flags)
            oe.close()


def test_search_reproduces_engine_py_trees():
    recs = fx.mcts_fixtures()
    assert len(recs) == 48
    terminal_hits = 0
    for rec in recs:
        oe = orc.Engine(fx.config_for(rec["fen"], rec["visits"]))
        fx.drive(oe.select, oe.leaf_boards, oe.backup, 1 + rec["visits"])
        s = oe.game_state(0)
        assert s.ply == 0 and s.phase == 2  # the move is due, not yet played
        fx.check_search(rec, oe.tree(0), s)
        terminal_hits += oe.stats()["steps"] - (oe.stats()["nn_evals"] - 1)
    assert terminal_hits > 0  # some searches met finished positions inside the tree


def test_per_ply_moves_messages_never_reuse_the_tree():
    # engine.set_state (engine.py:452-472) looks for a grand-child; uai_ringmaster.py sends one `moves` message per ply,
    # so the board it gets is a child: every move starts from an empty tree == AZH_FLAG_NO_REUSE
    with open(os.path.join(GOLDEN, "engine_reuse.json")) as f:
        plies = json.load(f)
    assert len(plies) == 12 and all(p["inherited_root_visits"] == 0 and p["root_visits_after"] == 40 for p in plies)


def test_training_samples_equal_train_py():
    with open(os.path.join(GOLDEN, "train_entries.json")) as f:
        entries = json.load(f)
    want = np.load(os.path.join(GOLDEN, "train_samples.npz"))
    kinds = set()
    for seed in range(64):
        random.seed(seed)
        f, p, v = training.get_sample_from_entries(entries)
        assert (np.asarray(f) == want["features"][seed]).all(), seed
        assert np.array_equal(np.asarray(p, dtype=np.float32), want["policy"][seed]), seed
        assert list(v) == list(want["value"][seed])
        kinds.add(int(np.count_nonzero(p)) > 1)
    assert kinds == {True, False}  # dists targets and one-hot targets both occur
    for group in json.loads(str(want["sym_moves"])):
        for s, src, dst in group:
            got = training.apply_symmetry_to_move(s, training.uai_decode_move(src))
            assert got == training.uai_decode_move(dst)


def test_symmetry_average_equals_nn_evals_evaluate():
    want = np.load(os.path.join(GOLDEN, "nn_evals_sym.npz"))
    feats = want["features"].astype(np.float64)
    for i, f in enumerate(feats):
        for s in range(8):
            assert (net_oracle.apply_symmetry(f, s) == want["images"][i, s]).all()
    assert [net_oracle.INVERSE_SYMMETRY[s] for s in range(8)] == want["inverse"].tolist()
    ev32 = lambda images: tuple(np.asarray(a, dtype=np.float32) for a in linear_evals(images))
    p, v = net_oracle.sym_average(ev32, feats)
    assert np.abs(p - want["policy"]).max() <= 1e-6 and np.abs(v.ravel() - want["value"]).max() <= 1e-6
    plain, _ = ev32(feats)
    assert np.abs(plain - want["policy"]).max() > 1e-2  # the average differs from the plain evaluation


def test_tree_reuse_across_moves_equals_mcts_play():
    # engine.MCTS.play (engine.py:411-424) keeps the chosen child's subtree; the searches continue from its counts
    recs = fx.reuse_fixtures()
    assert len(recs) == 16
    inherited = 0
    for rec in recs:
        oe = orc.Engine(fx.reuse_config(rec))
        fx.check_reuse_sequence(rec, oe.select, oe.leaf_boards, oe.backup, lambda: oe.game_state(0), lambda: oe.tree(0))
        st = oe.stats()
        assert st["reroot_nodes"] > len(rec["plies"])   # subtrees were kept, not rebuilt
        inherited += sum(p["root_visits"] - p["steps"] for p in rec["plies"][1:])
    assert inherited > 1000


def test_pgn_blocks_equal_the_reference_ringmaster(tmp_path):
    # arena.write_game_to_pgn against uai_ringmaster.write_game_to_pgn (uai_ringmaster.py:162-180): tag for tag, except
    # the three lines that carry wall-clock times
    from ataxxzero_amd import arena
    with open(os.path.join(GOLDEN, "ringmaster_pgn.json")) as f:
        fixture = json.load(f)
    clock = ("[Date ", "[GameStartTime ", "[GameEndTime ")
    for i, g in enumerate(fixture["games"]):
        path = str(tmp_path / ("g%d.pgn" % i))
        arena.write_game_to_pgn(path, {"moves": g["moves"], "result": g["result"], "final_score": g["final_score"]},
                                g["white"], g["black"], i + 1, fixture["tc"])
        got = [l for l in open(path).read().split("\n") if not l.startswith(clock)]
        want = [l for l in g["pgn"].split("\n") if not l.startswith(clock)]
        assert got == want
        assert sum(l.startswith(clock) for l in open(path).read().split("\n")) == 3


def _reference_random_games():
    import gzip
    with gzip.open(os.path.join(GOLDEN, "random_play_games.jsonl.gz")) as f:
        return [l for l in f.read().decode().split("\n") if l.strip()]


def _shape(v):
    """Structure of a JSON value: types and nesting, lists summarised by the set of their element shapes."""
    if isinstance(v, dict):
        return {k: _shape(x) for k, x in v.items()}
    if isinstance(v, list):
        return ["list", sorted({json.dumps(_shape(x), sort_keys=True) for x in v})]
    return type(v).__name__


def test_reference_random_play_games_replay_through_the_oracle_rules():
    """Six whole games written by the reference's own generate_games.py --random-play: every recorded board, every move
    and the result must be what the oracle's rules produce — movegen, makemove with captures and adjudication checked
    along complete reference games, in the reference's own file format."""
    from ataxxzero_amd import selfplay
    from tests.helpers import replay_game_entry
    lines = _reference_random_games()
    assert len(lines) == 6
    for line in lines:
        entry = json.loads(line)
        assert list(entry) == ["boards", "moves", "result"] and entry["result"] in (1, 2)
        sq = lambda xy: "abcdefg"[xy[0]] + str(7 - xy[1])
        uai = [sq(m[1]) if m[0] == "c" else sq(m[0]) + sq(m[1]) for m in entry["moves"]]
        assert replay_game_entry({"boards": entry["boards"], "moves": uai}, orc.START_FEN_PLAIN) == entry["result"]
        # the product's encoder writes the reference's move values: u16 move -> nested lists
        for m, text in zip(entry["moves"], uai):
            assert selfplay.python_move(orc.move_from_string(text)) == m


def test_random_play_entry_format_is_the_reference_format():
    # keys, their order, nesting and element types of a reference line (generate_games.py:50-51,:68,:134-136) against an
    # entry assembled by the product's own encoders (the GPU CLI is compared the same way in tests/test_gpu_cli.py)
    from ataxxzero_amd import selfplay
    ref = json.loads(_reference_random_games()[0])
    p = orc.pos_from_fen(orc.START_FEN_PLAIN)
    moves = orc.movegen(p)
    ours = {"boards": [selfplay.board_cells(int(p.pieces[0]), int(p.pieces[1]))] * 2,
            "moves": [selfplay.python_move(int(moves[0])), selfplay.python_move(int(moves[-1]))], "result": 2}
    assert {moves[0] & 0xFF == moves[0] >> 8, moves[-1] & 0xFF == moves[-1] >> 8} == {True, False}   # a jump and a clone
    assert _shape(ours) == _shape(ref) and list(ours) == list(ref)
    assert ours["boards"][0] == ref["boards"][0]                      # both start from ataxx_rules.AtaxxState.initial()
