#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Python engine (engine.py, train.py, nn_evals.py).

Run in the build container only (needs /root/reference, which never travels):

    PYTHONHASHSEED=0 python tests/golden/gen_engine_fixtures.py

The three modules import TensorFlow at module scope but the functions pinned here never touch it
(only `import tensorflow` and `tf.nn.relu` are evaluated at import, engine.py:5,
model.py:5,15), so an empty in-memory module named `tensorflow` with an `nn.relu` attribute is registered in
sys.modules before the import — nothing of TensorFlow is restated, and the reference files are imported
unmodified from where they lie.  The net itself (model.py's graph) is NOT run: the evaluator is injected
through the two module globals engine.py reads (`engine.sess`, `engine.network`, engine.py:184-190) as a pure
function of the feature rows — tests/helpers.synthetic_evals_distinct, the same function the tests feed to
the oracle and to the HIP engine.

Writes data-only fixtures next to this script:

  engine_features.npz     engine.board_to_features (engine.py:53-73) for every position of
                          rules_noblock.json.gz / rules_block4.json.gz (same order), int8 (n,7,7,4)
  engine_policy_index.json  flat index into the (7,7,17) posterior that engine.get_move_score
                          (engine.py:98-110) / add_move_to_heatmap (:80-87) use, for every clone and every jump
  engine_posterior.json.gz  NNEvaluator's posterior (engine.py:197-203) and value for 96 positions
  engine_dirichlet.json.gz  engine.add_dirichlet_noise_to_posterior (engine.py:117-124; the mix of
                          cpp/self_play_client.cpp:250-271) on 32 of those posteriors, alpha 0.15 / weight 0.25, with
                          np.random.dirichlet returning a recorded vector: the normalised gamma draws the search engines
                          make for that root (seed, uid 0, ply 0, edge j — oracle/detmath.h's gamma, in float64)
  engine_mcts.json.gz     48 deterministic searches driven exactly as uai_interface.py:44-46,79 drives them
                          (MCTSEngine.set_state + genmove(1e6, use_weighted_exponent=5.0) with MAX_STEPS = visits):
                          every edge of the final tree (path, visits, total score), root posterior, the move
                          weights of sample_with_exponential_weight (engine.py:532-548), smallest PUCT margin met
  engine_reuse_search.json.gz  16 multi-ply searches WITH tree reuse (MCTS.play, engine.py:411-424) whose every move is
                          forced (one root edge with n >= max/2): per ply the tree before the move, then the kept subtree
  engine_reuse.json       a 12-ply game played the way uai_ringmaster.py drives two engines (one `moves` message per
                          ply -> set_state): root visits found after every set_state (tree reuse never happens)
  train_samples.npz       train.get_sample_from_entries (train.py:43-77) under random.seed(k) on
                          train_entries.json (a small games file in both entry flavours)
  random_play_games.jsonl.gz  six games of generate_games.generate_game --random-play (generate_games.py:16-75), as written
  ringmaster_pgn.json     uai_ringmaster.write_game_to_pgn (uai_ringmaster.py:162-180) on three game dicts + the win tally
  nn_evals_sym.npz        nn_evals.evaluate (nn_evals.py:48-62) with tests/helpers.linear_evals injected
"""
import array
import gzip
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

_tf = types.ModuleType("tensorflow")
_tf.nn = types.SimpleNamespace(relu=None)
sys.modules["tensorflow"] = _tf

import ataxx_rules  # noqa: E402
import engine  # noqa: E402
import nn_evals  # noqa: E402
import train  # noqa: E402
import uai_interface  # noqa: E402

from tests import helpers  # noqa: E402

BLOCK4 = frozenset([(3, 2), (2, 3), (4, 3), (3, 4)])
enc = uai_interface.uai_encode_move


def set_blockers(cells):
    ataxx_rules.BLOCKED_CELLS = frozenset(cells)
    ataxx_rules.LEGAL_SQUARE_COUNT = ataxx_rules.SIZE * ataxx_rules.SIZE - len(cells)


def dump_gz(name, obj):
    with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as f:
        f.write(json.dumps(obj, separators=(",", ":")).encode())


def load_gz(name):
    with gzip.open(os.path.join(HERE, name)) as f:
        return json.loads(f.read())


# ------------------------------------------------------------------ injected evaluator (engine.py:184-190)

class FakeSession:
    """Stands where engine.sess (a tf.InteractiveSession) stands: run([policy, value], feed_dict)."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = 0

    def run(self, fetches, feed_dict):
        feats = np.asarray(feed_dict["input_ph"])
        self.calls += 1
        return self.fn(feats)


def distinct_on_features(feats):
    logits, values = helpers.synthetic_evals_distinct(helpers.leaf_boards_from_features(feats))
    return logits.reshape(-1, 7, 7, 17), values.reshape(-1, 1)


def install(fn):
    engine.network = types.SimpleNamespace(policy_output="policy", value_output="value", input_ph="input_ph",
                                           is_training_ph="is_training_ph")
    engine.sess = FakeSession(fn)
    engine.setup_evaluator(use_rpc=False, temperature=0.0)  # uai_interface.py:105


# ------------------------------------------------------------------ features / indices / posterior

def features_fixture():
    out = {}
    for name, cells in (("noblock", frozenset()), ("block4", BLOCK4)):
        set_blockers(cells)
        rows = []
        for rec in load_gz("rules_%s.json.gz" % name):
            # from the recorded cells: AtaxxState.from_fen does not read the '-' of a blocked cell
            state = ataxx_rules.AtaxxState(array.array("b", rec["cells"]), to_move=rec["to_move"])
            rows.append(engine.board_to_features(state))
        out[name] = np.asarray(rows, dtype=np.int8)
    set_blockers(frozenset())
    np.savez_compressed(os.path.join(HERE, "engine_features.npz"), **out)
    return {k: v.shape for k, v in out.items()}


def policy_index_fixture():
    probe = np.arange(7 * 7 * 17, dtype=np.float64).reshape(7, 7, 17)
    table = {}
    for x in range(7):
        for y in range(7):
            m = ("c", (x, y))
            table[enc(m)] = int(engine.get_move_score(probe, m))
            heat = engine.encode_move_as_heatmap(m)
            assert int(np.flatnonzero(heat.ravel())[0]) == table[enc(m)]
            for dx, dy in ataxx_rules.FAR_NEIGHBOR_OFFSETS:
                if 0 <= x + dx < 7 and 0 <= y + dy < 7:
                    m = ((x, y), (x + dx, y + dy))
                    table[enc(m)] = int(engine.get_move_score(probe, m))
                    heat = engine.encode_move_as_heatmap(m)
                    assert int(np.flatnonzero(heat.ravel())[0]) == table[enc(m)]
    with open(os.path.join(HERE, "engine_policy_index.json"), "w") as f:
        json.dump(table, f, sort_keys=True)
    return len(table)


def sample_positions(seed, n_games, per_game):
    """Non-terminal positions from seeded random games (no blockers: engine.py plays the plain board)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n_games):
        state = ataxx_rules.AtaxxState.initial()
        line = [state.copy()]
        while state.result() is None and len(line) < 400:
            state.move(rng.choice(state.legal_moves()))
            line.append(state.copy())
        live = [s for s in line if s.result() is None]
        # spread over the game, always including late positions (terminal nodes inside the tree)
        picks = sorted(set([0, len(live) - 1, len(live) - 3] + [rng.randrange(len(live)) for _ in range(per_game)]))
        out += [live[i] for i in picks if 0 <= i < len(live)]
    return out


def posterior_fixture():
    install(distinct_on_features)
    recs = []
    for state in sample_positions(99, 8, 10)[:96]:
        b = state.copy()
        engine.global_evaluator.populate(b)
        ev = b.evaluations
        recs.append({"fen": b.fen(), "value": float(ev.value),
                     "posterior": [[enc(m), float(p)] for m, p in ev.posterior.items()]})
    dump_gz("engine_posterior.json.gz", recs)
    return len(recs)


def dirichlet_fixture():
    """The root-noise mix.  The reference's C++ generator draws gamma(0.15) per root move, normalises and mixes
    0.75 p + 0.25 noise (cpp/self_play_client.cpp:250-271); engine.py states the same mix as
    add_dirichlet_noise_to_posterior (engine.py:117-124) and is importable.  Its random ingredient is replaced by a
    recorded vector — the draws the engines under test make for this root, normalised here in float64 — so what the
    fixture pins is the formula and which noise element meets which move."""
    from oracle import oracle_lib as orc
    install(distinct_on_features)
    alpha, weight = 0.15, 0.25
    recs = []
    real_dirichlet = np.random.dirichlet
    try:
        for i, state in enumerate(sample_positions(99, 8, 10)[:96:3]):
            b = state.copy()
            engine.global_evaluator.populate(b)
            posterior = b.evaluations.posterior                    # python legal_moves() order
            seed = 5000 + i
            order = [orc.move_string(m) for m in orc.movegen(orc.pos_from_fen(b.fen()))]   # the engines' edge order
            assert sorted(order) == sorted(enc(m) for m in posterior)
            gam = np.array([orc.lib().orc_probe_gamma(alpha, seed, 0, 0, j) for j in range(len(order))], dtype=np.float64)
            noise_of = dict(zip(order, gam / gam.sum()))
            vector = np.array([noise_of[enc(m)] for m in posterior])
            np.random.dirichlet = lambda alphas, v=vector: (v if len(alphas) == len(v) and set(alphas) == {alpha} else None)
            try:
                mixed = engine.add_dirichlet_noise_to_posterior(posterior, alpha, weight)
            except AssertionError:
                # the reference's own sanity check (engine.py:124): with the synthetic evaluator a position with two or
                # three legal moves can hold so little of the 833-way softmax mass that the 1e-6 of engine.py:202
                # shows in the sum; such a position is left out
                continue
            recs.append({"fen": b.fen(), "seed": seed, "alpha": alpha, "weight": weight,
                         "noise": [[m, float(noise_of[m])] for m in order],
                         "mixed": [[enc(m), float(p)] for m, p in mixed.items()]})
    finally:
        np.random.dirichlet = real_dirichlet
    dump_gz("engine_dirichlet.json.gz", recs)
    return len(recs)


# ------------------------------------------------------------------ searches

class MarginProbe:
    """Wraps MCTSNode.select_action: the reference's own answer is returned untouched; beside it the gap between the
    best and second-best total_action_score is noted, so a fixture says how close the search came to a tie."""

    def __init__(self):
        self.orig = engine.MCTSNode.select_action
        self.min_margin = float("inf")
        probe = self

        def wrapped(node, use_dirichlet_noise):
            move = probe.orig(node, use_dirichlet_noise)
            if move is not None:
                scores = sorted((node.total_action_score(m) for m in node.board.evaluations.posterior), reverse=True)
                if len(scores) > 1:
                    probe.min_margin = min(probe.min_margin, scores[0] - scores[1])
            return move

        engine.MCTSNode.select_action = wrapped

    def close(self):
        engine.MCTSNode.select_action = self.orig


def tree_edges(root):
    out = []
    frontier = [((), root)]
    while frontier:
        nxt = []
        for path, node in frontier:
            for move, edge in node.outgoing_edges.items():
                p = path + (enc(move),)
                out.append([" ".join(p), int(edge.edge_visits), float(edge.edge_total_score)])
                nxt.append((p, edge.child_node))
        frontier = nxt
    return out


def run_search(state, visits, seed):
    """Exactly what `uai_interface.py --visits V` does for `position fen ...` + `go movetime ...` (:44-46,:63-79)."""
    eng = engine.MCTSEngine()
    eng.MAX_STEPS = visits
    eng.set_state(state.copy())
    captured = {}
    orig_sample = engine.sample_by_weight

    def capture(weights):
        captured["weights"] = {enc(m): float(w) for m, w in weights.items()}
        return orig_sample(weights)

    engine.sample_by_weight = capture
    probe = MarginProbe()
    random.seed(seed)
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")  # genmove prints "info speed ..."
    try:
        move = eng.genmove(1000000.0, use_weighted_exponent=5.0)
    finally:
        sys.stdout.close()
        sys.stdout = stdout
        probe.close()
        engine.sample_by_weight = orig_sample
    root = eng.mcts.root_node
    ev = root.board.evaluations
    return {
        "fen": state.fen(), "visits": visits, "bestmove": enc(move),
        "root_visits": int(root.all_edge_visits), "root_value": float(ev.value),
        "root_posterior": [[enc(m), float(p)] for m, p in ev.posterior.items()],
        "edges": tree_edges(root), "move_weights": captured.get("weights", {}),
        "min_margin": probe.min_margin,
    }


def mcts_fixture():
    install(distinct_on_features)
    positions = sample_positions(2026, 6, 6)
    rng = random.Random(5)
    rng.shuffle(positions)
    recs = []
    for i, state in enumerate(positions[:48]):
        visits = (50, 100, 200, 120)[i % 4]
        recs.append(run_search(state, visits, seed=i))
    dump_gz("engine_mcts.json.gz", recs)
    return recs


def reuse_fixture():
    """Two engines fed one `moves` message per ply, as uai_ringmaster.py:123-134 does: what set_state (engine.py:452-472)
    leaves as the root's visit count.  0 at every ply = the tree is rebuilt every time."""
    install(distinct_on_features)
    board = ataxx_rules.AtaxxState.initial()
    engs = [engine.MCTSEngine(), engine.MCTSEngine()]
    for e in engs:
        e.MAX_STEPS = 40
    random.seed(3)
    plies = []
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        for ply in range(12):
            e = engs[ply % 2]
            inherited = int(e.mcts.root_node.all_edge_visits)
            move = e.genmove(1000000.0, use_weighted_exponent=5.0)
            plies.append({"fen": board.fen(), "inherited_root_visits": inherited, "move": enc(move),
                          "root_visits_after": int(e.mcts.root_node.all_edge_visits)})
            board.move(move)
            for other in engs:
                other.set_state(board.copy())  # uai_interface.py:58-62
    finally:
        sys.stdout.close()
        sys.stdout = stdout
    with open(os.path.join(HERE, "engine_reuse.json"), "w") as f:
        json.dump(plies, f)
    return plies


def reuse_search_fixture():
    """Multi-ply searches WITH tree reuse, pinned by the reference's own MCTS.play (engine.py:411-424): search until the
    root has `visits` visits (the C++ generator's rule, cpp/self_play_client.cpp:522 — inherited visits count), take the
    move sample_with_exponential_weight would return, MCTS.play it (the chosen child's subtree and counts are kept),
    search again.  Only sequences in which every move is FORCED are kept: exactly one root edge has n >= max/2, so the
    exponent-5 sample is that move whatever the random number — the engine under test draws from Philox, the
    reference from `random`, and neither matters."""
    install(distinct_on_features)
    recs = []
    positions = sample_positions(31337, 30, 5)
    rng = random.Random(9)
    rng.shuffle(positions)
    for state in positions:
        if len(recs) >= 16:
            break
        visits = (60, 100, 150)[len(recs) % 3]
        mcts = engine.MCTS(state.copy())
        plies = []
        ok = True
        for ply in range(4):
            steps = 0
            while mcts.root_node.all_edge_visits < visits:
                if mcts.step() is None:   # terminal root
                    ok = False
                    break
                steps += 1
            if not ok:
                break
            root = mcts.root_node
            top = max(e.edge_visits for e in root.outgoing_edges.values())
            forced = [m for m, e in root.outgoing_edges.items() if e.edge_visits >= top * 0.5]
            if len(forced) != 1:
                ok = False
                break
            move = forced[0]
            plies.append({"fen": root.board.fen(), "steps": steps, "root_visits": int(root.all_edge_visits),
                          "move": enc(move), "edges": tree_edges(root)})
            mcts.play(root.board.to_move, move)
            if mcts.root_node.board.result() is not None:
                break
        if ok and len(plies) >= 3:
            recs.append({"fen": state.fen(), "visits": visits, "plies": plies,
                         "kept_root_visits": int(mcts.root_node.all_edge_visits), "kept_edges": tree_edges(mcts.root_node)})
    dump_gz("engine_reuse_search.json.gz", recs)
    return recs


# ------------------------------------------------------------------ train.py sample pipeline

def train_fixture():
    rng = random.Random(11)
    entries = []
    for k in range(6):
        state = ataxx_rules.AtaxxState.initial()
        boards, moves, dists = [], [], []
        while state.result() is None and len(moves) < 400:
            legal = state.legal_moves()
            m = rng.choice(legal)
            boards.append(list(state.board))
            moves.append(m)
            picks = rng.sample(legal, min(3, len(legal)))
            if m not in picks:
                picks[0] = m
            w = [rng.randrange(1, 20) for _ in picks]
            dists.append({enc(p): wi / sum(w) for p, wi in zip(picks, w)})
            state.move(m)
        result = state.result()
        if k % 3 == 0:   # C++ generator flavour: UAI strings + dists (cpp/self_play_client.cpp:565-578)
            entries.append({"boards": boards, "dists": dists, "moves": [enc(m) for m in moves], "result": result})
        elif k % 3 == 1:  # python generator flavour: nested lists, no dists (generate_games.py:50-51)
            entries.append({"boards": boards, "moves": [[m[0] if m[0] == "c" else list(m[0]), list(m[1])] for m in moves],
                            "result": result})
        else:             # ONE_RANDOM_MOVE flavour (train.py:47-49)
            entries.append({"boards": boards, "dists": dists, "moves": [enc(m) for m in moves], "result": result,
                            "random_ply": rng.randrange(0, len(moves) - 1)})
    with open(os.path.join(HERE, "train_entries.json"), "w") as f:
        json.dump(entries, f, separators=(",", ":"))
    # json round trip first: the consumer sees lists, not tuples (train.py:79-89)
    entries = json.loads(json.dumps(entries))
    feats, pols, vals = [], [], []
    for seed in range(64):
        random.seed(seed)
        f, p, v = train.get_sample_from_entries(entries)
        feats.append(f)
        pols.append(p)
        vals.append(v)
    sym_moves = [[[s, enc(m), enc(train.apply_symmetry_to_move(s, uai_interface.uai_decode_move(enc(m))))]
                  for s in range(8)] for m in (("c", (1, 5)), ((0, 2), (2, 3)), ((6, 6), (4, 6)))]
    np.savez_compressed(os.path.join(HERE, "train_samples.npz"), features=np.asarray(feats, dtype=np.int8),
                        policy=np.asarray(pols, dtype=np.float32), value=np.asarray(vals, dtype=np.int8),
                        sym_moves=np.asarray(json.dumps(sym_moves)))
    return len(feats)


# ------------------------------------------------------------------ nn_evals.evaluate

def sym_fixture():
    install(lambda feats: tuple(np.asarray(a, dtype=np.float32) for a in helpers.linear_evals(feats)))
    boards = sample_positions(7, 2, 4)[:8]
    feats = np.asarray([engine.board_to_features(b) for b in boards], dtype=np.int8)
    pol, val = [], []
    for b in boards:
        p, v = nn_evals.evaluate(b)
        pol.append(p)
        val.append(v)
    images = np.asarray([[nn_evals.apply_symmetry(f, s) for s in range(8)] for f in feats], dtype=np.int8)
    np.savez_compressed(os.path.join(HERE, "nn_evals_sym.npz"), features=feats, images=images,
                        policy=np.asarray(pol, dtype=np.float64), value=np.asarray(val, dtype=np.float64),
                        inverse=np.asarray([nn_evals.inverse_symmetry[s] for s in range(8)], dtype=np.int8))
    return len(boards)


def pgn_fixture():
    """uai_ringmaster.write_game_to_pgn (uai_ringmaster.py:162-180) on three hand-made game dicts, and the scoring rule
    of its main loop (:251-257) applied to them."""
    import argparse
    import tempfile
    import uai_ringmaster
    dec = uai_interface.uai_decode_move
    games = [
        {"white": ["python", "uai_interface.py", "--network-path", "a.npy", "--visits", "100"],
         "black": ["python", "uai_interface.py", "--network-path", "b.npy", "--visits", "100"],
         "opening": [], "start_time": 1700000000.0, "end_time": 1700000042.5,
         "moves": [dec(m) for m in ("a7b6", "g7f6", "b6", "f5")], "result": 1, "final_score": (30, 19)},
        {"white": ["engine", "two"], "black": ["engine", "one"], "opening": [], "start_time": 1700000100.0,
         "end_time": 1700000101.0, "moves": [dec(m) for m in ("g1", "a1a3")], "result": 2, "final_score": (0, 7)},
        {"white": ["x"], "black": ["y"], "opening": [], "start_time": 1700000200.0, "end_time": 1700000300.0,
         "moves": [dec("b7")] * 3, "result": "invalid", "final_score": (5, 5)},
    ]
    args = argparse.Namespace(tc=1.0)
    texts = []
    for i, g in enumerate(games):
        with tempfile.NamedTemporaryFile("r", suffix=".pgn") as f:
            uai_ringmaster.write_game_to_pgn(args, f.name, g, round_index=i + 1)
            texts.append(open(f.name).read())
    wins = {"white": 0, "black": 0}
    annulled = 0
    for g in games:   # uai_ringmaster.py:251-257 (both sides are scored per colour here)
        if g["result"] in (1, 2):
            wins[("white", "black")[g["result"] - 1]] += 1
        else:
            wins["white"] += 0.5
            wins["black"] += 0.5
            annulled += 1
    out = [{"white": " ".join(g["white"]), "black": " ".join(g["black"]), "moves": [enc(m) for m in g["moves"]],
            "result": 0 if g["result"] == "invalid" else g["result"], "final_score": list(g["final_score"]), "pgn": t}
           for g, t in zip(games, texts)]
    with open(os.path.join(HERE, "ringmaster_pgn.json"), "w") as f:
        json.dump({"tc": 1.0, "games": out, "wins": wins, "annulled": annulled}, f)
    return len(out)


def random_play_fixture():
    """generate_games.generate_game with --random-play (generate_games.py:16-75), six games under random.seed(0..5), each
    dumped the way the script's write loop dumps it (json.dump, default separators, :134-136)."""
    import argparse
    import io
    import generate_games
    args = argparse.Namespace(random_play=True, supervised=None, show_game=False, die_if_present=None, group_index=0,
                              visit_count=0)
    lines = []
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        for seed in range(6):
            random.seed(seed)
            entry = generate_games.generate_game(args)
            buf = io.StringIO()
            json.dump(entry, buf)
            lines.append(buf.getvalue())
    finally:
        sys.stdout.close()
        sys.stdout = stdout
    with gzip.GzipFile(os.path.join(HERE, "random_play_games.jsonl.gz"), "wb", mtime=0) as f:
        f.write(("\n".join(lines) + "\n").encode())
    return [len(json.loads(l)["moves"]) for l in lines]


def main():
    if os.environ.get("PYTHONHASHSEED") != "0":
        print("note: run with PYTHONHASHSEED=0 for byte-stable output (clone moves come out of a set, "
              "ataxx_rules.py:137-153)", file=sys.stderr)
    print("features:", features_fixture())
    print("policy indices:", policy_index_fixture())
    print("posteriors:", posterior_fixture())
    print("dirichlet mixes:", dirichlet_fixture())
    recs = mcts_fixture()
    print("searches:", len(recs), "edges:", sum(len(r["edges"]) for r in recs),
          "min margin: %.3g" % min(r["min_margin"] for r in recs))
    rr = reuse_search_fixture()
    print("reuse searches:", len(rr), "plies:", [len(r["plies"]) for r in rr],
          "inherited visits:", [p["root_visits"] - p["steps"] for r in rr for p in r["plies"][1:]][:12])
    plies = reuse_fixture()
    print("reuse: inherited root visits per ply:", [p["inherited_root_visits"] for p in plies])
    print("train samples:", train_fixture())
    print("sym boards:", sym_fixture())
    print("pgn games:", pgn_fixture())
    print("random-play games (plies):", random_play_fixture())


if __name__ == "__main__":
    main()
