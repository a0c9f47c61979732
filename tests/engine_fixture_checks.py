"""Shared checks that replay tests/golden/engine_*.json.gz — vectors produced by the reference's own engine.py
(tests/golden/gen_engine_fixtures.py) — against a search engine with the oracle's / the HIP engine's Python surface
(select / leaf boards / backup / tree).  The engine under test runs with the Python-engine flags
(AZH_FLAG_NO_REUSE | TIE_FIRST | PY_POSTERIOR | SAMPLE_POW5), no Dirichlet noise (engine.py:325), one game slot whose
start position is the fixture's position and whose visit threshold is the fixture's step count."""
import numpy as np

from oracle import oracle_lib as orc
from tests.helpers import load_gz, synthetic_evals_distinct

PY_ENGINE_FLAGS = orc.FLAG_NO_REUSE | orc.FLAG_TIE_FIRST | orc.FLAG_PY_POSTERIOR | orc.FLAG_SAMPLE_POW5
NONE = 0xFFFFFFFF


def mcts_fixtures():
    return load_gz("engine_mcts.json.gz")


def posterior_fixtures():
    return load_gz("engine_posterior.json.gz")


def config_for(fen, visits):
    return orc.make_config(games=1, visits=visits, seed=1, fen_str=fen, max_plies=400, weight=0.0,
                           flags=PY_ENGINE_FLAGS)


def dirichlet_fixtures():
    return load_gz("engine_dirichlet.json.gz")


def dirichlet_config(rec, flags):
    """One game slot at the fixture's position with the fixture's seed: uid 0, ply 0 — the key of the recorded draws."""
    return orc.make_config(games=1, visits=4, seed=rec["seed"], fen_str=rec["fen"], max_plies=400, alpha=rec["alpha"],
                           weight=rec["weight"], flags=flags)


def check_dirichlet(rec, root, flags):
    """Root priors after the first backup against engine.add_dirichlet_noise_to_posterior's float64 mix
    (engine.py:117-124 == cpp/self_play_client.cpp:250-271).  With the python posterior underneath the engine and the
    fixture compute the same expression: f32 rounding only.  The C++ generator's posterior (softmax renormalised over
    the legal moves, no 1e-6 in the denominator) differs from engine.py's by the factor 1 + 1e-6 / legal mass, so
    there the (1 - w) p part is compared after that factor is taken out."""
    got = dict(root)
    want = dict(rec["mixed"])
    noise = dict(rec["noise"])
    assert set(got) == set(want) == set(noise) and list(got) == [m for m, _ in rec["noise"]]
    w = rec["weight"]
    scale = 1.0
    if not flags & orc.FLAG_PY_POSTERIOR:
        scale = (1.0 - w) / sum(want[m] - w * noise[m] for m in want)   # python posterior sums to 1 / (1 + 1e-6 / mass)
    for m in want:
        ref = w * noise[m] + scale * (want[m] - w * noise[m])
        assert abs(got[m] - ref) <= 4e-6 * ref + 1e-12, (m, got[m], ref)
    assert abs(sum(got.values()) - (1.0 if scale != 1.0 else sum(want.values()))) < 2e-5


def walk_tree(tree):
    """(boards, info, edges, moves) arena dump -> {path of UAI moves: (visits, total score, prior)} for expanded edges,
    and the root's [(uai, prior)] in movegen order."""
    boards, info, edges, moves = tree
    out = {}
    frontier = [((), 0)]
    while frontier:
        nxt = []
        for path, node in frontier:
            first, m = int(info[node, 0]), int(info[node, 1] & 0xFFFF)
            for j in range(first, first + m):
                if int(edges[j, 3]) != NONE:
                    p = path + (orc.move_string(int(moves[j])),)
                    out[" ".join(p)] = (int(edges[j, 1]), float(edges[j, 2:3].copy().view(np.float32)[0]),
                                        float(edges[j, 0:1].copy().view(np.float32)[0]))
                    nxt.append((p, int(edges[j, 3])))
        frontier = nxt
    first, m = int(info[0, 0]), int(info[0, 1] & 0xFFFF)
    root = [(orc.move_string(int(moves[j])), float(edges[j, 0:1].copy().view(np.float32)[0]))
            for j in range(first, first + m)]
    return out, root


def check_priors(root, want):
    """root priors (f32) against engine.py:197-203's float64 posterior."""
    got = dict(root)
    assert set(got) == {m for m, _ in want}
    for m, p in want:
        assert abs(got[m] - p) <= 3e-6 * p + 1e-12, (m, got[m], p)


def check_search(rec, tree, state):
    """The finished search against the reference's tree: same expanded edges, identical visit counts, total scores
    to f32 accumulation error, priors to f32 rounding, and the (n/N)^5 move weights of engine.py:532-548."""
    got, root = walk_tree(tree)
    assert state.root_visits == rec["root_visits"] == rec["visits"]
    check_priors(root, rec["root_posterior"])
    want = {p: (n, w) for p, n, w in rec["edges"]}
    assert set(got) == set(want), (sorted(set(got) ^ set(want))[:5], rec["fen"], rec["min_margin"])
    for p, (n, w) in want.items():
        gn, gw, _ = got[p]
        assert gn == n, (p, gn, n, rec["fen"])
        assert abs(gw - w) <= n * 2e-7 * max(1.0, w), (p, gw, w)
    # sample_with_exponential_weight (engine.py:532-548): weights (n/N)^5 over root edges with n >= max/2,
    # normalised — the engine samples from the exact integers n^5 over the same support
    visits = {p: n for p, (n, _, _) in got.items() if " " not in p}
    top = max(visits.values())
    w5 = {m: float(n) ** 5 for m, n in visits.items() if 2 * n >= top}
    total = sum(w5.values())
    assert set(w5) == set(rec["move_weights"])
    for m, w in rec["move_weights"].items():
        assert abs(w5[m] / total - w) <= 1e-12 + 1e-9 * w


def drive(engine_select, engine_leaves, engine_backup, iterations):
    for _ in range(iterations):
        engine_select()
        logits, values = synthetic_evals_distinct(engine_leaves())
        engine_backup(logits, values)


# ------------------------------------------------------------------ multi-ply searches with tree reuse

REUSE_FLAGS = orc.FLAG_TIE_FIRST | orc.FLAG_PY_POSTERIOR | orc.FLAG_SAMPLE_POW5   # python engine, trees KEPT across moves


def reuse_fixtures():
    return load_gz("engine_reuse_search.json.gz")


def reuse_config(rec):
    return orc.make_config(games=1, visits=rec["visits"], seed=3, fen_str=rec["fen"], max_plies=400, weight=0.0,
                           flags=REUSE_FLAGS)


def check_edges(tree, want_edges):
    got, _ = walk_tree(tree)
    want = {p: (n, w) for p, n, w in want_edges}
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:5]
    for p, (n, w) in want.items():
        assert got[p][0] == n, (p, got[p][0], n)
        assert abs(got[p][1] - w) <= n * 2e-7 * max(1.0, w), (p, got[p][1], w)


def check_reuse_sequence(rec, select, leaves, backup, game_state, tree):
    """Drive one game slot through the fixture's plies: before every move the tree must be the tree engine.py had when
    its root reached `visits` visits (inherited visits included), the forced move must be the move played, and after
    MCTS.play the kept subtree must be the reference's, edge for edge."""
    ply = checked = 0
    for _ in range(rec["visits"] * len(rec["plies"]) + 50):
        select()
        s = game_state()
        if s.uid != 0:
            # the last forced move ended the game (the slot has already restarted): every ply was compared, and the
            # reference's kept root is that finished position, with nothing under it
            assert checked == len(rec["plies"]) and rec["kept_edges"] == []
            return
        if s.ply > ply:   # this select played the move that was due
            ply = s.ply
            if ply == len(rec["plies"]):
                assert s.root_visits == rec["kept_root_visits"]
                check_edges(tree(), rec["kept_edges"])
                return
        logits, values = synthetic_evals_distinct(leaves())
        backup(logits, values)
        s = game_state()
        if s.phase == 2:   # the move is due: the finished search of this ply
            want = rec["plies"][s.ply]
            assert s.root_visits == want["root_visits"], (s.ply, s.root_visits, want["root_visits"])
            check_edges(tree(), want["edges"])
            checked += 1
    raise AssertionError("the fixture's plies were not all played")
