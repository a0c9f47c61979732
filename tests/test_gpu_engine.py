"""Search parity: the HIP engine against the CPU oracle (oracle/mcts_oracle.c), fed the
same evaluator outputs, compared bit for bit: per-game state, whole tree arenas
(boards, edge priors / visits / scores / children as raw 32-bit patterns), finished
game records."""
import json
import os

import numpy as np
import pytest

from ataxxzero_amd import link, model, selfplay
from oracle import oracle_lib as orc
from tests.helpers import replay_game_entry, synthetic_evals

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_pair(games, visits, max_plies=400, edges_per_node=96, seed=77, fen=orc.START_FEN_SELFPLAY, weight=0.25,
              flags=0, select_budget=0):
    ocfg = orc.make_config(games, visits, seed=seed, fen_str=fen, max_plies=max_plies,
                           edges_per_node=edges_per_node, weight=weight, flags=flags, select_budget=select_budget)
    gcfg = link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_})
    return orc.Engine(ocfg), link.Engine(gcfg)


def _is_nan_bits(w):
    return ((w & 0x7F800000) == 0x7F800000) & ((w & 0x007FFFFF) != 0)


def compare_all(oe, ge, games, nan_aware=False):
    """nan_aware: a NaN prior or total score must be a NaN on both sides, whatever its sign and payload — the only words of
    a tree that are not a function of IEEE arithmetic alone (which NaN an operation hands on is the hardware's choice)."""
    for g in games:
        so, sg = oe.game_state(g), ge.game_state(g)
        assert so.as_tuple() == sg.as_tuple(), (g, so.as_tuple(), sg.as_tuple())
        to, tg = oe.tree(g), ge.tree(g)
        for name, a, b in zip(("boards", "info", "edges", "moves"), to, tg):
            assert a.shape == b.shape, (g, name)
            if nan_aware and name == "edges":
                a, b = a.copy(), b.copy()
                for col in (0, 2):
                    na, nb = _is_nan_bits(a[:, col]), _is_nan_bits(b[:, col])
                    assert (na == nb).all(), (g, name, col)
                    a[na, col] = b[nb, col] = 0x7FC00000
            assert (a == b).all(), (g, name)


def check_marks(ge, games):
    """The early request's mark (bit 31 of a prior in the device's edge records, azh_engine_tree_raw): at most ONE marked edge
    per node, and only on an edge whose child exists, is not a finished position and has 1 .. 128 moves — what lets the
    descent request the marked child's records without a check.  -> (nodes with a mark, marks on an edge index >= 64)."""
    marked_nodes = high = 0
    for g in games:
        _, info, _, _ = ge.tree(g)
        raw = ge.tree_raw(g)
        mark = (raw[:, 0] >> 31) != 0
        for n in range(len(info)):
            first, m = int(info[n, 0]), int(info[n, 1] & 0xFFFF)
            idx = np.nonzero(mark[first:first + m])[0]
            assert len(idx) <= 1, (g, n, idx.tolist())
            if len(idx):
                assert m <= 128, (g, n, m)
                z, w = int(raw[first + idx[0], 2]), int(raw[first + idx[0], 3])
                assert (z >> 16) != 0xFFFF and (w >> 31) == 0 and 1 <= ((w >> 23) & 0xFF) <= 128, (g, n, int(idx[0]), hex(z), hex(w))
                marked_nodes += 1
                high += int(idx[0]) >= 64
        # no mark outside the nodes' ranges either (every edge belongs to exactly one node)
        assert int(mark.sum()) <= len(info)
    return marked_nodes, high


def run_lockstep(oe, ge, iterations, check_every=1, evaluator=synthetic_evals):
    G = oe.G
    o_games, g_lines = [], []
    for it in range(iterations):
        n_o, need_o = oe.select()
        n_g = ge.select()
        need_g, lb_g = ge.leaves()
        lb_o = oe.leaf_boards()
        assert n_o == n_g and (need_o == need_g).all(), it
        assert (lb_o[need_o == 1] == lb_g[need_o == 1]).all(), it
        logits, values = evaluator(lb_o)
        oe.backup(logits, values)
        ge.set_evals(logits, values)
        ge.backup()
        if it % check_every == 0 or it == iterations - 1:
            compare_all(oe, ge, range(G))
        o_games += oe.pop_games()
        g_lines += ge.drain_json()
    return o_games, g_lines


def test_engine_matches_oracle_bit_for_bit_small_visits():
    oe, ge = make_pair(games=24, visits=12, max_plies=60, seed=5)
    o_games, g_lines = run_lockstep(oe, ge, 900, check_every=7)
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert so["plies"] > 200 and so["games"] + so["dropped"] > 3
    o_sorted = sorted(o_games, key=lambda r: r["uid"])
    assert len(g_lines) == len(o_sorted)
    for line, rec in zip(g_lines, o_sorted):
        entry = json.loads(line)
        assert list(entry.keys()) == ["boards", "dists", "moves", "result"]  # nlohmann: sorted keys
        assert entry == rec["entry"]
        assert b" " not in line
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


def test_engine_matches_oracle_reference_settings():
    # the reference's constants: 4-blocker start, Dirichlet 0.15 / 0.25, c = 1
    oe, ge = make_pair(games=6, visits=100, seed=20260101)
    run_lockstep(oe, ge, 450, check_every=50)
    assert oe.stats()["plies"] >= 6


def test_edge_arena_overflow_forces_the_move_like_the_oracle():
    oe, ge = make_pair(games=4, visits=64, edges_per_node=8, seed=9, fen=orc.START_FEN_PLAIN)
    run_lockstep(oe, ge, 300, check_every=10)
    assert oe.stats()["edge_overflow"] > 0 and ge.stats()["edge_overflow"] == oe.stats()["edge_overflow"]


def test_marks_of_the_early_request_in_nodes_with_more_than_64_moves():
    """Mid-game positions (a third of their nodes have 65-128 moves: two edges per lane) searched in lock step with the
    oracle, then the raw marks.  Round 5 moved a mark with ONE store in which a lane chose between setting and clearing: when
    the newly chosen edge and the previously marked one were 64 apart — the two edges of one lane — the clear was dropped
    and the node kept two marks.  The evaluator here makes that case the rule: wherever a position has more than 67 moves
    its 4th and its 68th move (movegen order) get the same, dominant prior, so PUCT alternates between two edges that are
    exactly 64 apart.  (With round 5's store this test fails on its first check — two marks, on edges 3 and 67 of one node:
    profiles/round6_mark_collision_test.txt.)"""
    snap = np.load(os.path.join(ROOT, "profiles", "round2_steady_state_positions.npz"))
    rng = np.random.default_rng(3)
    cand = np.nonzero((snap["plies"] > 40) & (snap["plies"] < 200))[0]
    pick = rng.choice(cand, size=48, replace=False)
    oe, ge = make_pair(games=48, visits=300, seed=6)
    blockers = oe.cfg.blockers
    forced = [0]

    def evaluator(lb):
        logits, values = synthetic_evals(lb)
        for i in range(len(lb)):
            p = orc.Pos()
            p.pieces[0], p.pieces[1], p.blockers, p.turn = int(lb[i, 0]), int(lb[i, 1]), blockers, 0   # the mover's moves
            moves = orc.movegen(p)
            if len(moves) > 67:
                logits[i] = -6.0
                logits[i, orc.lib().orc_policy_index(int(moves[3]))] = 3.0
                logits[i, orc.lib().orc_policy_index(int(moves[67]))] = 3.0
                forced[0] += 1
        return logits, values

    for e in (oe, ge):
        e.set_positions(snap["boards"][pick], snap["plies"][pick])
    run_lockstep(oe, ge, 290, check_every=97, evaluator=evaluator)
    big = sum(int(((ge.tree(g)[1][:, 1] & 0xFFFF) > 64).sum()) for g in range(48))
    marked, high = check_marks(ge, range(48))
    assert forced[0] > 500 and big > 300 and marked > 1000 and high > 20, (forced[0], big, marked, high)


def test_nan_evaluations_are_never_selected_and_never_read_as_a_mark():
    """NaN from the evaluator (both references never select a NaN score, cpp/self_play_client.cpp:345-358, engine.py:291).
    NaN or infinite LOGITS never reach a prior: exp() of the deterministic f32 library returns 0 for them, a row of them
    gives all-zero priors on both sides — so bit 31 of a stored prior, the descent's mark, is never a NaN's sign (and
    priors are stored without a sign in any case).  A NaN VALUE makes the total score of every edge on its path NaN: such
    edges are never selected again, a node whose edges are all NaN falls back to its first move — in lock step with the
    oracle, marks intact."""
    neg_nan = np.array([0xFFC00001], dtype=np.uint32).view(np.float32)[0]

    def evaluator(lb):
        logits, values = synthetic_evals(lb)
        logits[(lb[:, 0] % np.uint64(5)) == 0] = neg_nan
        logits[(lb[:, 1] % np.uint64(7)) == 0, 100:300] = -neg_nan     # part of a row, the other sign
        logits[(lb[:, 1] % np.uint64(11)) == 0, ::3] = np.inf
        values[(lb[:, 0] % np.uint64(41)) == 0] = neg_nan
        return logits, values

    oe, ge = make_pair(games=40, visits=60, max_plies=120, seed=15)
    G = oe.G
    nan_scores = zero_rows = 0
    for it in range(1500):
        n_o, need_o = oe.select()
        assert ge.select() == n_o
        logits, values = evaluator(oe.leaf_boards())
        oe.backup(logits, values)
        ge.set_evals(logits, values)
        ge.backup()
        if it % 250 == 249:
            compare_all(oe, ge, range(G), nan_aware=True)
            check_marks(ge, range(G))
            for g in range(G):
                _, info, edges, _ = ge.tree(g)
                assert not _is_nan_bits(edges[:, 0]).any() and not (edges[:, 0] >> 31).any()
                nan_scores += int(_is_nan_bits(edges[:, 2]).sum())
                zero_rows += sum(1 for n in range(len(info)) if (info[n, 1] & 0xFFFF) and (info[n, 1] >> 16) == 0 and
                                 n > 0 and not edges[int(info[n, 0]):int(info[n, 0]) + int(info[n, 1] & 0xFFFF), 0].any())
        oe.pop_games(), ge.drain_json()
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert nan_scores > 50 and zero_rows > 50 and so["plies"] > 40, (nan_scores, zero_rows, so["plies"])


def test_leaf_features_are_reference_rows():
    oe, ge = make_pair(games=16, visits=8, seed=3)
    for _ in range(5):
        n_o, need = oe.select()
        n = ge.select()
        feats, games = ge.leaf_features(n)
        assert list(games) == [g for g in range(16) if need[g]]
        for row, g in zip(feats, games):
            assert (row == oe.leaf_features(int(g))).all()
        logits, values = synthetic_evals(oe.leaf_boards())
        oe.backup(logits, values)
        ge.set_evals(logits, values)
        ge.backup()


@pytest.mark.parametrize("dtype,visits", [("bf16", 24), ("f16", 100)])
def test_device_resident_selfplay_with_builtin_net(dtype, visits):
    conv, bn = model.random_init(2, 128, seed=2)
    net = link.Net(conv, bn)
    ocfg = orc.make_config(64, visits, seed=11, max_plies=400)
    ge = link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))
    lines = []
    for _ in range(10 * visits):
        ge.run(net, 200, link.DTYPES[dtype])
        lines += ge.drain_json()
        if len(lines) >= 40:
            break
    st = ge.stats()
    assert st["games"] == len(lines) >= 40 and st["ring_overflow"] == 0 and st["edge_overflow"] == 0
    assert st["nn_evals"] > st["plies"] and st["steps"] >= st["nn_evals"] - st["plies"] - 64
    for line in lines[:20]:
        entry = json.loads(line)
        assert entry["result"] in (1, 2)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]
        # root visit counts reach the threshold: each dist is k / N with N >= visits
        assert all(min(d.values()) > 0 for d in entry["dists"])


def test_overlapped_round_trip_never_waits_for_the_device_inside_the_drain():
    """The generator's order — fetch, enqueue the next run, drain (accelerated_generate_games.py) — on the real library: no
    call of the drain sequence may fetch by itself (it would wait for the run just enqueued, and the round would be
    sequential again without saying so: round 3's loop did that), the drain must come back while that run is still in
    flight, and the lines are the ones the sequential order gives."""
    import time
    conv, bn = model.random_init(2, 128, seed=2)
    net = link.Net(conv, bn)
    ocfg = orc.make_config(256, 24, seed=11, max_plies=400)

    def engine():
        return link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))

    seq, ovl = engine(), engine()
    want, seq_rounds = [], []
    for _ in range(12):
        seq.run(net, 300, link.DTYPE_F32)
        seq_rounds.append(seq.drain_json())
        want += seq_rounds[-1]
    assert seq.implicit_fetches() == 12          # a caller that never fetches: one fetch per drain sequence, not two
    lines, drain_s, rest_s, empty_rounds = [], 0.0, 0.0, 0
    ovl.run(net, 300, link.DTYPE_F32)
    for _ in range(11):
        ovl.fetch()
        ovl.run(net, 300, link.DTYPE_F32)
        t0 = time.perf_counter()
        got = ovl.drain_json()
        t1 = time.perf_counter()
        ovl.sync()                                # what is left of the run that was enqueued before the drain
        rest_s += time.perf_counter() - t1
        drain_s += t1 - t0
        empty_rounds += not got
        lines += got
    ovl.fetch()
    lines += ovl.drain_json()
    assert ovl.implicit_fetches() == 0
    assert lines == want and len(lines) > 100
    # (a note, not an assertion — implicit_fetches() == 0 pins the property: had the drain waited for the device, nothing of
    # the run would be left after it)
    print("overlapped drain: %.1f ms formatting in all under %.1f ms of search left after it, %d of 11 rounds without a game"
          % (1e3 * drain_s, 1e3 * rest_s, empty_rounds))
    # a C caller that never fetches and drains with ONE call per round (a buffer large enough for everything): every round
    # gets its own fetch and its own games — none arrives a round late
    import ctypes
    one = engine()
    buf = np.zeros(1 << 26, dtype=np.uint8)
    got = []
    for _ in range(12):
        one.run(net, 300, link.DTYPE_F32)
        used, n = ctypes.c_int64(0), ctypes.c_int32(0)
        link.check(link.load().azh_engine_drain_json(one.h, ctypes.c_void_p(buf.ctypes.data), buf.nbytes, ctypes.byref(used),
                                                     ctypes.byref(n)))
        chunk = bytes(buf[:used.value]).split(b"\n")[:-1]
        assert len(chunk) == n.value
        got.append(chunk)
    assert one.implicit_fetches() == 12 and sum(got, []) == want
    assert [len(c) for c in got] == [len(c) for c in seq_rounds]
    # two engines on one GPU (the half-batches): a fetch of one waits for ITS OWN streams only and copies through pinned
    # staging — it comes back while the other engine's long run is still in flight (a stream query, not a clock)
    quick, slow = engine(), engine()
    quick.run(net, 100, link.DTYPE_F32)
    quick.sync()
    slow.run(net, 6000, link.DTYPE_F32)        # about two seconds of search
    quick.run(net, 100, link.DTYPE_F32)
    quick.fetch()
    still_running = slow.busy()
    quick.drain_json()
    slow.sync()
    assert still_running and not slow.busy() and not quick.busy()


def test_game_limit_plays_exactly_the_games_below_it_and_then_idles():
    """azh_engine_set_game_limit: uids 0 .. N - 1 are played (slot g: g, g + G, ...), a slot whose next game would be past
    the limit goes idle, and the games are the ones the unlimited engine plays under those uids."""
    conv, bn = model.random_init(1, 128, seed=9)
    net = link.Net(conv, bn)
    G, N = 48, 100
    ocfg = orc.make_config(G, 6, seed=3, max_plies=400)
    mk = lambda: link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))
    limited, free = mk(), mk()
    limited.set_emit_order(True)
    free.set_emit_order(True)
    limited.set_game_limit(N)
    lines = []
    for _ in range(400):
        limited.run(net, 100, link.DTYPE_F32)
        lines += limited.drain_json()
        st = limited.stats()
        if st["games"] + st["dropped"] >= N:
            break
    assert st["games"] + st["dropped"] == N and len(lines) == st["games"]
    assert all(limited.game_state(g).phase == 3 and limited.game_state(g).uid >= N for g in range(G))
    limited.run(net, 50, link.DTYPE_F32)
    limited.sync()
    assert limited.stats() == st and limited.drain_json() == []     # idle: no search, no games
    # the unlimited engine, same seed: its first games in uid order are the same games (f32 tower: position-independent)
    want = []
    for _ in range(400):
        free.run(net, 100, link.DTYPE_F32)
        want += free.drain_json()
        if len(want) >= len(lines):
            break
    assert want[:len(lines)] == lines
    with pytest.raises(link.AzhError):
        limited.set_game_limit(0)


def test_game_limit_matches_oracle_in_lockstep():
    # the same rule in the oracle (orc_engine_set_game_limit): slots idle at the same iteration, states and trees equal
    oe, ge = make_pair(games=12, visits=6, max_plies=90, seed=4)
    oe.set_game_limit(30)
    ge.set_game_limit(30)
    o_games, g_lines = run_lockstep(oe, ge, 2500, check_every=37)
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert so["games"] + so["dropped"] == 30 and so["dropped"] > 0 and len(g_lines) == len(o_games) == so["games"]
    assert all(ge.game_state(g).phase == 3 for g in range(12))
    # the limit is raised by what the drops left missing: the idle slots start those games, in both engines alike
    more = so["dropped"]
    oe.set_game_limit(30 + more)
    ge.set_game_limit(30 + more)
    assert sorted(ge.game_state(g).uid for g in range(12) if ge.game_state(g).phase != 3) == list(range(30, 30 + more))[:12]
    o2, g2 = run_lockstep(oe, ge, 2500, check_every=41)
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert so["games"] + so["dropped"] == 30 + more and len(g2) == len(o2) and all(ge.game_state(g).phase == 3 for g in range(12))


def test_reroot_queue_spill_path_matches_oracle():
    # visits > 512: kept subtrees grow past the LDS part of the re-root frontier queue, so the
    # HBM spill path of advance_game is exercised; still bit-exact against the oracle
    # (a steep prior over the policy index concentrates the visits on one line, so most of the tree is kept)
    steep = lambda lb: (np.tile(-0.75 * np.arange(833, dtype=np.float32), (len(lb), 1)), np.zeros(len(lb), np.float32))
    oe, ge = make_pair(games=2, visits=1200, seed=13, weight=0.0)
    run_lockstep(oe, ge, 3700, check_every=600, evaluator=steep)
    st = oe.stats()
    assert st["plies"] >= 4 and st["reroot_nodes"] / st["plies"] > 520, st


def test_odd_sizes_and_f16_device_loop():
    conv, bn = model.random_init(1, 128, seed=4)
    net = link.Net(conv, bn)
    for games, visits in ((1, 5), (67, 9)):
        ocfg = orc.make_config(games, visits, seed=2)
        ge = link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))
        ge.run(net, 150, link.DTYPE_F16)
        ge.sync()
        st = ge.stats()
        assert st["steps"] > 40 * games and st["plies"] > 5 * games and st["edge_overflow"] == 0
        ge.close()


def test_one_random_move_variant_matches_oracle():
    # cpp/self_play_client.cpp:515-552 (ONE_RANDOM_MOVE build): entries carry "random_ply"
    oe, ge = make_pair(games=32, visits=8, max_plies=240, seed=11, flags=orc.FLAG_ONE_RANDOM_MOVE)
    assert link.FLAG_ONE_RANDOM_MOVE == orc.FLAG_ONE_RANDOM_MOVE
    o_games, g_lines = run_lockstep(oe, ge, 2400, check_every=31)
    o_sorted = sorted(o_games, key=lambda r: r["uid"])
    assert len(g_lines) == len(o_sorted) and len(g_lines) >= 4
    # "Skipping game with no board state just after the uniformly random move" (cpp/self_play_client.cpp:632-637):
    # such games are dropped, not written, so train.py's `ply = random_ply + 1` always indexes a recorded board
    assert all(r["entry"]["random_ply"] + 1 < len(r["entry"]["moves"]) for r in o_sorted)
    assert oe.stats()["dropped"] == ge.stats()["dropped"] > 0
    from ataxxzero_amd import training
    feats, pols, vals = training.make_minibatch([json.loads(l) for l in g_lines], 64)
    assert feats.shape == (64, 7, 7, 4) and np.allclose(pols.sum(axis=(1, 2, 3)), 1, atol=1e-3)
    canon = lambda en: json.dumps(en, sort_keys=True)
    # games finish in different iterations, so compare as sets (uids are not part of the line)
    assert sorted(canon(json.loads(l)) for l in g_lines) == sorted(canon(r["entry"]) for r in o_sorted)
    for line in g_lines:
        entry = json.loads(line)
        assert list(entry.keys()) == ["boards", "dists", "moves", "random_ply", "result"]
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


@pytest.mark.parametrize("dtype,games", [("f32", 9), ("bf16", 96)])
def test_symmetry_averaging_flag_in_the_device_loop_equals_host_evaluated_search(dtype, games):
    # AZH_FLAG_SYMMETRY_AVG: every evaluation of the device loop is nn_evals.evaluate; the same search
    # driven from the host with Net.forward_sym must build bit-identical trees — f32, and bf16 with the three-board tower
    # (96 games x 8 images: the leaf list in game order puts every virtual board in the slot it has in the loop's launch)
    dt = link.DTYPES[dtype]
    conv, bn = model.random_init(1, 128, seed=31)
    net = link.Net(conv, bn)
    base = orc.make_config(games, 12, seed=4, max_plies=80, flags=link.FLAG_SYMMETRY_AVG)
    mk = lambda: link.Engine(link.Config(**{n: getattr(base, n) for n, _ in orc.Config._fields_}))
    dev, host = mk(), mk()
    if dtype != "f32":
        dev.set_thin_batches(0)          # (azh_net_forward_sym runs the three-board kernel)
    iters = 90
    dev.run(net, iters, dt)
    dev.sync()
    for _ in range(iters):
        host.select()
        need, lb = host.leaves()
        logits = np.zeros((games, 833), np.float32)
        values = np.zeros(games, np.float32)
        idx = np.nonzero(need)[0]
        if len(idx):
            p, v = net.forward_sym(lb[idx], base.blockers, dt)
            logits[idx] = p.reshape(len(idx), 833)
            values[idx] = v.reshape(-1)
        host.set_evals(logits, values)
        host.backup()
    for g in range(games):
        assert dev.game_state(g).as_tuple()[:6] == host.game_state(g).as_tuple()[:6]
        for a, b in zip(dev.tree(g), host.tree(g)):
            assert a.shape == b.shape and (a == b).all()
    assert dev.stats()["plies"] == host.stats()["plies"] > games


def test_parked_descents_match_oracle_and_leave_every_game_unchanged():
    # select_budget: a descent deeper than the budget parks and resumes next iteration.  Lockstep parity with the
    # oracle's same rule, and the games written are exactly the games of the unbudgeted engine.
    oe, ge = make_pair(games=16, visits=12, max_plies=300, seed=21, select_budget=2)
    o_games, g_lines = run_lockstep(oe, ge, 3000, check_every=13)
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert len(g_lines) >= 3
    parked = 0
    for _ in range(30):
        ge.select()
        parked += sum(ge.game_state(g).leaf_kind == 4 for g in range(16))
        need, lb = ge.leaves()
        lg, v = synthetic_evals(lb)
        ge.set_evals(lg, v)
        ge.backup()
    assert parked > 0  # the budget really bites
    # same seeds without a budget: the same games, bit for bit (only the iteration they finish in differs)
    _, ge0 = make_pair(games=16, visits=12, max_plies=300, seed=21)
    want = len(g_lines)
    lines0 = []
    for _ in range(4000):
        ge0.select()
        need, lb = ge0.leaves()
        lg, v = synthetic_evals(lb)
        ge0.set_evals(lg, v)
        ge0.backup()
        lines0 += ge0.drain_json()
        if len(lines0) >= want + 16:
            break
    assert set(g_lines) <= set(lines0)


@pytest.mark.parametrize("name,visits,blocks,net_seed,dtype,flags,streams", [
    ("C3-shard", 400, 12, 1, "bf16", 0, 1),            # BASELINE configs[2]'s per-GPU shard as one batch
    ("C3-shard-two-halves", 400, 12, 1, "bf16", 0, 2),  # ... as bench.py and the generator run it: two half-batches in flight
    ("C2", 200, 12, 1, "bf16", 0, 1),                  # configs[1]: 4096 games, 200 sims, 12x128, bf16
    ("C4", 800, 8, 3, "f16", 0, 1),                    # configs[3]: 8x128, fp16, 800 sims (node_cap 808)
    ("C3-shard-cached", 400, 12, 1, "bf16", link.FLAG_EVAL_CACHE, 1),   # the generator CLI's default mode
])
def test_full_size_workload_invariants(name, visits, blocks, net_seed, dtype, flags, streams):
    """BASELINE's full sizes (4096 games; 200 / 400 / 800 sims/move; 12x128 bf16 and 8x128 f16; level budget 48 or 64)
    are checked through size-independent properties of the search: tree bookkeeping identities on sampled games, counter
    identities over the whole batch, and replay of the games written.  `streams` = 2 is selfplay.SelfPlay's double
    buffer (two engines of 2048 games enqueued together, cpp/self_play_client.cpp:593-600): the configuration
    bench.py's `value` is measured on."""
    conv, bn = model.random_init(blocks, 128, seed=net_seed)
    G, V = 4096, visits
    sp = selfplay.SelfPlay(conv, bn, games=G, visits=V, dtype=dtype, seed=20260101, streams=streams,
                           select_budget=64 if flags else 48, flags=flags)
    assert all(e.node_cap == V + 8 for e in sp.engines) and sum(e.G for e in sp.engines) == G
    sp.set_visits(16)
    lines = []
    for _ in range(6):
        sp.run(250)            # (every engine's whole run is enqueued before any of them is waited for)
        lines += sp.drain()
    sp.set_visits(V)
    for _ in range(4):
        sp.run(250)
        lines += sp.drain()
    sp.sync()
    st = sp.stats()
    assert st["edge_overflow"] == 0 and st["ring_overflow"] == 0
    assert st["games"] == len(lines) > 500
    assert (st["cache_hits"] > 0) == bool(flags) and st["parked"] > 0
    # every step is one descent that ends in an evaluation or a terminal re-hit; evaluations = steps that reached a new
    # non-terminal node + one root (re-)evaluation per ply and per game start
    assert st["steps"] + st["parked"] <= 2500 * G
    assert st["nn_evals"] + st["cache_hits"] <= st["steps"] + st["plies"] + st["games"] + st["dropped"] + G
    assert st["levels"] >= st["steps"]  # every descent looks at the root at least
    rng = np.random.default_rng(0)
    for gg in rng.choice(G, size=48, replace=False):
        ge, g = sp.engines[0], int(gg)
        for e in sp.engines:   # global slot -> (engine, its slot)
            if g < e.G:
                ge = e
                break
            g -= e.G
        s = ge.game_state(g)
        boards, info, edges, moves = ge.tree(g)
        assert 1 <= s.n_nodes <= ge.node_cap and s.n_edges <= ge.edge_cap
        first, m = int(info[0, 0]), int(info[0, 1] & 0xFFFF)
        assert int(edges[first:first + m, 1].sum()) == s.root_visits
        kids = edges[:, 3][edges[:, 3] != 0xFFFFFFFF]
        assert len(set(kids.tolist())) == len(kids) == s.n_nodes - 1  # a tree: every node but the root has one parent
        n = edges[:, 1].astype(np.float64)
        w = edges[:, 2].copy().view(np.float32).astype(np.float64)
        assert ((w >= -1e-6) & (w <= n + 1e-3)).all()  # W / n is a probability of winning
        # visits of an edge = 1 (the expansion) + visits below it, for every expanded non-terminal child
        # (a parked descent has not backed up yet, so the identity is exact at every sync point)
        for e_idx in np.nonzero(edges[:, 3] != 0xFFFFFFFF)[0][:64]:
            c = int(edges[e_idx, 3])
            cf, cm, cres = int(info[c, 0]), int(info[c, 1] & 0xFFFF), int(info[c, 1] >> 16)
            if cres == 0 and cm > 0:
                assert int(edges[e_idx, 1]) == 1 + int(edges[cf:cf + cm, 1].sum()), (int(gg), int(e_idx))
    for line in lines[:: max(1, len(lines) // 200)]:
        entry = json.loads(line)
        assert entry["result"] in (1, 2)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]
    sp.close()


def _oracle_follow(oe, net, blockers, iterations, dtype=link.DTYPE_F32, thin=False, net_b=None):
    """The oracle plays `iterations` search iterations, its leaves evaluated by the tower the device-resident loop uses for
    them — the f32 tower, or a 16-bit tower with one board per workgroup (`thin`): both are bit-identical wherever a board
    sits in a launch (tests/test_gpu_net.py) — i.e. exactly what the loop computes for itself.  `net_b`: the arena's
    second net, for the leaves whose mover it is (need class 2, uai_ringmaster.py:221-265)."""
    G = oe.G
    logits = np.zeros((G, 833), np.float32)
    values = np.zeros(G, np.float32)
    for _ in range(iterations):
        _, need = oe.select()
        lb = None
        for cls, n in ((1, net), (2, net_b)):
            idx = np.nonzero(need == cls)[0] if net_b is not None else (np.nonzero(need)[0] if cls == 1 else [])
            if len(idx):
                lb = oe.leaf_boards() if lb is None else lb
                p, v = n.forward(lb[idx], blockers, dtype, thin=thin)
                logits[idx] = p.reshape(len(idx), 833)
                values[idx] = v.reshape(-1)
        oe.backup(logits, values)


@pytest.mark.parametrize("name,games,visits,blocks,max_plies,budget,chunks,chunk,dtype", [
    ("turnover", 512, 12, 2, 90, 6, 12, 250, "f32"),        # many plies: games finish, are cut at max_plies, slots restart
    ("budget48", 512, 100, 2, 400, 48, 6, 100, "f32"),      # the bench's level budget
    ("C2-shape", 768, 200, 12, 400, 48, 4, 150, "bf16"),    # BASELINE configs[1]: 12x128 net, 200 sims/move, bf16 (3-board tower)
    ("C4-shape", 576, 800, 8, 400, 48, 4, 450, "f16"),      # BASELINE configs[3]: 8x128 net, 800 sims/move, f16, node_cap 808
    ("bench-size", 4096, 400, 12, 400, 48, 3, 150, "bf16"), # bench.py's workload as one batch: 4096 games, 400 sims/move, 12x128, bf16
    ("C2-full", 4096, 200, 12, 400, 48, 3, 150, "bf16"),    # BASELINE configs[1] at its full size: 4096 games, 200 sims/move, bf16
    ("C4-full", 4096, 800, 8, 400, 48, 2, 300, "f16"),      # BASELINE configs[3] at its full size: 4096 games, 800 sims/move, 8x128, f16
    ("two-rounds", 8203, 8, 1, 60, 6, 2, 40, "f32"),        # more games than resident waves (8192): one game per workgroup,
                                                            # a need-bit mask with a ragged last word
    ("turnover-side-stream", 512, 12, 2, 90, 6, 12, 250, "f32"),   # rounds 3-5's loop (AZH_REROOT_SIDE_STREAM=1): the queued moves
                                                                   # as k_advance_list on a side stream behind events
])
def test_device_resident_loop_matches_oracle_bit_for_bit(name, games, visits, blocks, max_plies, budget, chunks, chunk, dtype, monkeypatch):
    """The loop bench.py and the CLI run — azh_engine_run: fused k_tree, the queued moves played by the first workgroups of
    the tower launch, parked descents — against the oracle, iteration for iteration: every game state, every arena word and
    every JSON line.  (The step-wise API the other lock-step tests drive shares the device functions but not the launch
    structure.)  The two BASELINE shapes run in their own dtype with the three-board tower (more than 512 slots): the oracle's
    leaves are evaluated by the same kernel in leaf-list order, which gives every board the slot it has in the loop."""
    dt = link.DTYPES[dtype]
    if name.endswith("side-stream"):
        monkeypatch.setenv("AZH_REROOT_SIDE_STREAM", "1")
        name = name[:-len("-side-stream")]
    conv, bn = model.random_init(blocks, 128, seed=7)
    net = link.Net(conv, bn)
    oe, ge = make_pair(games=games, visits=visits, max_plies=max_plies, seed=99, select_budget=budget)
    assert ge.node_cap == visits + 8
    g_lines = []
    parked = 0
    if name in ("bench-size", "C2-full", "C4-full"):
        # bench.py's spread: games are taken off ply 0 at 16 sims/move first, then the trees regrow at full sims
        for e in (oe, ge):
            e.set_visits(16)
        ge.run(net, 500, dt)
        _oracle_follow(oe, net, oe.cfg.blockers, 500, dtype=dt)
        for e in (oe, ge):
            e.set_visits(visits)
    for c in range(chunks):
        ge.run(net, chunk, dt)
        _oracle_follow(oe, net, oe.cfg.blockers, chunk, dtype=dt)
        ge.sync()
        # (beyond 5000 games every fifth game and both ends: the tree dumps are one copy per array and game)
        sample = range(games) if games <= 5000 else sorted(set(range(0, games, 5)) | set(range(64)) | set(range(games - 64, games)))
        compare_all(oe, ge, sample)
        parked += sum(oe.game_state(g).leaf_kind == orc.LEAF_DESCENT for g in sample)
        o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])  # a drain hands its games out in uid order
        g_chunk = ge.drain_json()
        assert len(g_chunk) == len(o_chunk), c
        for line, rec in zip(g_chunk, o_chunk):
            assert json.loads(line) == rec["entry"]
        g_lines += g_chunk
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert sg["ring_overflow"] == 0 and so["plies"] >= games
    marked, _ = check_marks(ge, range(0, games, max(1, games // 256)))   # the early request's marks, as the loop left them
    assert marked > 0
    if name == "turnover":
        assert len(g_lines) > games // 8 and so["dropped"] > 0 and parked > 0
        assert so["reroot_nodes"] > so["plies"]  # subtrees are really kept across moves


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_half_batches_in_flight_match_their_oracles_at_bench_size(dtype):
    """bench.py's and the generator's default since round 4, at the size `value` is measured on: selfplay.SelfPlay(streams=2) =
    two engines of 2048 games, 400 sims/move, 12x128, level budget 48, sharing one packed weight set (the reference's double
    buffer, cpp/self_play_client.cpp:593-600; per game the loop of :419-473).  Both halves' runs are ENQUEUED before either
    is waited for — a stream each, their iterations enqueued in turn (azh_engines_run), each half's tree launches squeezed under the other
    half's tower — and each half is then compared with an oracle engine of its own: every game state, every arena word,
    every JSON line.  (The f32 tower is bit-identical wherever a board sits in a launch and whatever runs beside it; bf16 —
    the dtype and the kernel the headline is measured with — is followed with the oracle's leaves handed to the same
    3-board kernel in leaf-list order, i.e. in the slots they have in the loop's launches.)"""
    seed, G, V = 4242, 4096, 400
    dt = link.DTYPES[dtype]
    conv, bn = model.random_init(12, 128, seed=7)
    sp = selfplay.SelfPlay(conv, bn, games=G, visits=V, dtype=dtype, seed=seed, streams=2, select_budget=48)
    oes = [orc.Engine(orc.make_config(G // 2, V, seed=seed + 1000003 * i, select_budget=48)) for i in range(2)]
    for oe, ge in zip(oes, sp.engines):
        for n, _ in orc.Config._fields_:
            assert getattr(oe.cfg, n) == getattr(ge.cfg, n), n
    blockers = oes[0].cfg.blockers

    def advance(iterations):
        sp.run(iterations)                       # both halves enqueued, neither synced
        for oe in oes:
            _oracle_follow(oe, sp.net, blockers, iterations, dtype=dt)
        sp.sync()

    # bench.py's spread: games are taken off ply 0 at 16 sims/move first, then the trees regrow at full sims
    for e in oes + [sp]:
        e.set_visits(16)
    advance(500)
    for e in oes + [sp]:
        e.set_visits(V)
    n_lines = 0
    for c in range(2):
        advance(150)
        for oe, ge in zip(oes, sp.engines):
            compare_all(oe, ge, range(ge.G))
            o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])
            g_chunk = ge.drain_json()
            assert len(g_chunk) == len(o_chunk), c
            for line, rec in zip(g_chunk, o_chunk):
                assert json.loads(line) == rec["entry"]
            n_lines += len(g_chunk)
    for oe, ge in zip(oes, sp.engines):
        so, sg = oe.stats(), ge.stats()
        for k in so:
            assert so[k] == sg[k], (k, so[k], sg[k])
        assert sg["ring_overflow"] == 0 and so["plies"] >= ge.G and sg["parked"] > 0
    assert n_lines > 0
    sp.close()


def test_thin_batch_switch_in_the_device_loop():
    """azh_engine_set_thin_batches in the middle of a run (what arena.Match and the generator do for their last games): with
    the f32 tower, which has one kernel, the search stays the oracle's bit for bit whatever the mode; with bf16 the one-board-
    per-workgroup kernel takes over and the search stays a valid search (counters consistent, written games replay)."""
    conv, bn = model.random_init(2, 128, seed=7)
    net = link.Net(conv, bn)
    oe, ge = make_pair(games=600, visits=24, max_plies=120, seed=13, select_budget=8)   # 600 slots: the 3-board tower by default
    for mode in (-1, 1, 0):
        ge.set_thin_batches(mode)
        ge.run(net, 60, link.DTYPE_F32)
        _oracle_follow(oe, net, oe.cfg.blockers, 60)
        ge.sync()
        compare_all(oe, ge, range(0, 600, 7))
    with pytest.raises(link.AzhError):
        ge.set_thin_batches(2)
    sp = selfplay.SelfPlay(conv, bn, games=600, visits=24, dtype="bf16", seed=5, max_plies=120)
    lines = []
    for mode in (-1, 1, 1, 0, 1):
        sp.set_thin_batches(mode)
        sp.run(150)
        lines += sp.drain()
    st = sp.stats()
    assert st["games"] == len(lines) > 20 and st["edge_overflow"] == 0 and st["ring_overflow"] == 0
    assert st["nn_evals"] <= st["steps"] + st["plies"] + st["games"] + st["dropped"] + 600
    for line in lines[:40]:
        entry = json.loads(line)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]
    sp.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_thin_batch_device_loop_in_16_bits_matches_oracle_bit_for_bit(dtype):
    """The one-board-per-workgroup tower inside the device-resident loop (engines of at most 512 slots pick it by themselves;
    arena.Match and the generator switch to it for their last games) pinned to the oracle: the thin kernel is bit-identical
    wherever a board sits in a launch (tests/test_gpu_net.py), so an oracle whose leaves go through azh_net_forward_thin sees
    exactly the loop's evaluations — states, arena words, lines, counters.  Both ways of getting there: by size (384 slots)
    and by azh_engine_set_thin_batches(1) on a larger engine."""
    dt = link.DTYPES[dtype]
    conv, bn = model.random_init(3, 128, seed=17)
    net = link.Net(conv, bn)
    for games, force in ((384, False), (640, True)):
        oe, ge = make_pair(games=games, visits=20, max_plies=70, seed=23, select_budget=6)
        if force:
            ge.set_thin_batches(1)
        n_lines = 0
        for c in range(6):
            ge.run(net, 200, dt)
            _oracle_follow(oe, net, oe.cfg.blockers, 200, dtype=dt, thin=True)
            ge.sync()
            compare_all(oe, ge, range(games))
            o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])
            g_chunk = ge.drain_json()
            assert [json.loads(l) for l in g_chunk] == [r["entry"] for r in o_chunk], c
            n_lines += len(g_chunk)
        so, sg = oe.stats(), ge.stats()
        for k in so:
            assert so[k] == sg[k], (k, so[k], sg[k])
        assert n_lines > 20 and so["dropped"] > 0 and so["plies"] > 10 * games
        ge.close()


@pytest.mark.parametrize("dtype,streams,flags", [("bf16", 1, 0), ("f16", 1, 0), ("bf16", 2, 0),
                                                 ("bf16", 2, link.FLAG_EVAL_CACHE)])   # the last: the generator CLI's defaults
def test_three_board_16_bit_tower_in_the_device_loop_matches_oracle_bit_for_bit(dtype, streams, flags):
    """The loop as bench.py and the generator run it by default — the 16-bit tower with three boards per workgroup — pinned to
    the oracle bit for bit.  A board's last bits depend on its slot in its workgroup, and the slot on its place in the
    iteration's leaf list; but the leaf list is the games that need an evaluation in ascending order, so the oracle's leaves
    handed to azh_net_forward in that order sit in the same slots of the same kernel (a board's cells are its own columns of
    the contraction: what its neighbours in the workgroup hold does not matter).  `streams` = 2: two half-batches in flight,
    their runs enqueued in turn (azh_engines_run), each followed by an oracle of its own."""
    dt = link.DTYPES[dtype]
    seed, G, V = 515, 1536 * streams, 40
    conv, bn = model.random_init(4, 128, seed=19)
    sp = selfplay.SelfPlay(conv, bn, games=G, visits=V, dtype=dtype, seed=seed, streams=streams, select_budget=8, max_plies=110,
                           flags=flags)
    oes = [orc.Engine(orc.make_config(G // streams, V, seed=seed + 1000003 * i, select_budget=8, max_plies=110, flags=flags))
           for i in range(streams)]
    assert all(e.G > link.THIN_MAX_GAMES for e in sp.engines)      # the 3-board kernel, by the engines' size
    n_lines = 0
    for c in range(5):
        sp.run(220)
        for oe in oes:
            _oracle_follow(oe, sp.net, oe.cfg.blockers, 220, dtype=dt)
        sp.sync()
        for oe, ge in zip(oes, sp.engines):
            compare_all(oe, ge, range(0, ge.G, 3))
            o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])
            g_chunk = ge.drain_json()
            assert [json.loads(l) for l in g_chunk] == [r["entry"] for r in o_chunk], c
            n_lines += len(g_chunk)
    for oe, ge in zip(oes, sp.engines):
        so, sg = oe.stats(), ge.stats()
        for k in so:
            assert so[k] == sg[k], (k, so[k], sg[k])
        assert sg["parked"] > 0 and so["plies"] > 10 * ge.G and (sg["cache_hits"] > 0) == bool(flags)
    assert n_lines > 50
    sp.close()


def test_uid_ordered_emission_is_an_unbiased_prefix():
    """azh_engine_set_emit_order(1): games come out in the order they were STARTED.  What has been handed out at any
    moment is exactly the finished games among the uids below the smallest uid still in play — short and long games
    alike, dropped games (cut at max_plies) leaving no hole that blocks the queue."""
    oe, ge = make_pair(games=48, visits=6, max_plies=170, seed=17)   # the cut drops the longer half of the games
    ge.set_emit_order(True)
    o_games, g_lines = run_lockstep(oe, ge, 7000, check_every=1000)
    frontier = min(oe.game_state(g).uid for g in range(48))      # smallest uid not yet finished or dropped
    finished = sorted(o_games, key=lambda r: r["uid"])
    want = [r["entry"] for r in finished if r["uid"] < frontier]
    assert oe.stats()["dropped"] > 0 and len(finished) > len(want) > 48   # some games are held back, several generations out
    assert [json.loads(l) for l in g_lines] == want
    # finish order hands the same games out earlier, and a biased subset: shorter on average than the uid prefix
    lengths_all = [len(r["entry"]["moves"]) for r in finished]
    first_finishers = [len(r["entry"]["moves"]) for r in o_games[:len(want) // 2]]
    prefix = [len(e["moves"]) for e in want[:len(want) // 2]]
    assert np.mean(first_finishers) < np.mean(prefix) + 1e-9 and np.mean(first_finishers) < np.mean(lengths_all)


def test_lost_records_do_not_stall_uid_ordered_emission():
    """The record ring holds what finishes between two drains (64 KiB per slot, at least 16 MiB).  A host that does not
    drain for far too long loses records (AZH_STAT_RING_OVERFLOW counts them, the CLI exits non-zero on it).  In uid
    order a lost uid would be waited for for ever: instead the order is given up from that drain on — everything held
    is handed out and later games come as they arrive."""
    conv, bn = model.random_init(1, 128, seed=12)
    net = link.Net(conv, bn)
    ocfg = orc.make_config(256, 4, seed=6, max_plies=400)
    ge = link.Engine(link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_}))
    ge.set_emit_order(True)
    for _ in range(150):                      # ~3 k games of ~1.5 k words each without a single drain: the 4 M words overflow
        ge.run(net, 500, link.DTYPE_BF16)
        ge.sync()
        if ge.stats()["ring_overflow"] > 0:
            break
    st = ge.stats()
    assert st["ring_overflow"] > 0, st
    first = ge.drain_json()
    assert len(first) > 1000                  # what did fit comes out at once, gaps or not
    later = []
    for _ in range(6):
        ge.run(net, 300, link.DTYPE_BF16)
        later += ge.drain_json()
    st2 = ge.stats()
    # no stall: the games that finished after the overflow drain are all handed out
    assert st2["ring_overflow"] == st["ring_overflow"] and len(later) == st2["games"] - st["games"] > 100
    for line in (first[:5] + later[:5]):
        entry = json.loads(line)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


def test_loaded_positions_hook_matches_oracle():
    # azh_engine_set_positions (bench.py's steady-state set-up): every slot restarts at a mid-game position and ply; those
    # games are played and counted but not written; the slot's next game is an ordinary one
    oe, ge = make_pair(games=24, visits=10, max_plies=200, seed=8)
    run_lockstep(oe, ge, 700, check_every=100)                      # let the games reach different plies first
    boards = np.array([ge.tree(g)[0][0] for g in range(24)], dtype=np.uint64)
    plies = np.array([ge.game_state(g).ply for g in range(24)], dtype=np.int32)
    assert plies.max() > 20 and len(set(plies.tolist())) > 5
    oe2, ge2 = make_pair(games=24, visits=10, max_plies=200, seed=9)
    for e in (oe2, ge2):
        e.set_positions(boards, plies)
    for g in range(24):
        assert ge2.game_state(g).ply == plies[g] and (ge2.tree(g)[0][0] == boards[g]).all()
    o_games, g_lines = run_lockstep(oe2, ge2, 4000, check_every=250)
    so, sg = oe2.stats(), ge2.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    # the 24 loaded games ended (finished or cut) without a line; later games of the slots are written normally
    assert so["games"] + so["dropped"] > 24 and 0 < len(g_lines) == len(o_games) <= so["games"] + so["dropped"] - 24 + 24
    assert len(g_lines) < so["games"] + so["dropped"]
    for line in g_lines:
        entry = json.loads(line)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]   # complete games from the start position
    with pytest.raises(link.AzhError):
        ge2.set_positions(boards, np.full(24, 200, np.int32))       # ply beyond max_plies


def test_evaluation_cache_through_game_turnover_matches_oracle():
    # many plies, finished and cut games, slots restarting: the table is cleared / rebuilt at every one of those events
    conv, bn = model.random_init(2, 128, seed=7)
    net = link.Net(conv, bn)
    oe, ge = make_pair(games=384, visits=12, max_plies=90, seed=99, select_budget=6, flags=orc.FLAG_EVAL_CACHE)
    written = 0
    for c in range(10):
        ge.run(net, 250, link.DTYPE_F32)
        _oracle_follow(oe, net, oe.cfg.blockers, 250)
        ge.sync()
        compare_all(oe, ge, range(384))
        o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])
        g_chunk = ge.drain_json()
        assert [json.loads(l) for l in g_chunk] == [r["entry"] for r in o_chunk]
        written += len(g_chunk)
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert written > 40 and so["dropped"] > 0 and so["cache_hits"] > 0 and so["reroot_nodes"] > so["plies"]


def test_evaluation_cache_matches_oracle_and_saves_evaluations():
    """AZH_FLAG_EVAL_CACHE in the device-resident loop against the oracle's same rule, f32 tower on both sides (so an
    evaluation taken from the cache is bit for bit what the net would return): states, trees, game lines and counters;
    and against the uncached engine: the same games, fewer evaluations."""
    conv, bn = model.random_init(2, 128, seed=7)
    net = link.Net(conv, bn)
    oe, ge = make_pair(games=256, visits=60, max_plies=400, seed=31, select_budget=48, flags=orc.FLAG_EVAL_CACHE)
    assert link.FLAG_EVAL_CACHE == orc.FLAG_EVAL_CACHE
    lines = []
    for c in range(5):
        ge.run(net, 300, link.DTYPE_F32)
        _oracle_follow(oe, net, oe.cfg.blockers, 300)
        ge.sync()
        compare_all(oe, ge, range(256))
        o_chunk = sorted(oe.pop_games(), key=lambda r: r["uid"])
        g_chunk = ge.drain_json()
        assert [json.loads(l) for l in g_chunk] == [r["entry"] for r in o_chunk]
        lines += g_chunk
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    # (2 % here: 60 sims/move from the opening; about a quarter at 400 sims/move in mid-game, tools/leaf_duplicates.py)
    assert sg["cache_hits"] > 0.01 * sg["nn_evals"], (sg["cache_hits"], sg["nn_evals"])
    # the uncached engine plays the same games (same seeds), with more evaluations
    _, plain = make_pair(games=256, visits=60, max_plies=400, seed=31, select_budget=48)
    plain_lines = []
    for c in range(5):
        plain.run(net, 300, link.DTYPE_F32)
        plain_lines += plain.drain_json()
    sp = plain.stats()
    assert plain_lines == lines and sp["steps"] == sg["steps"] and sp["plies"] == sg["plies"]
    assert sp["nn_evals"] == sg["nn_evals"] + sg["cache_hits"]
