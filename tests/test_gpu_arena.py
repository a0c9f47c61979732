"""Arena (config 5): the HIP engine with the arena flags against the oracle, bit for bit,
plus the batched two-net match and the uai_ringmaster.py drop-in."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from ataxxzero_amd import arena, link, model
from oracle import oracle_lib as orc
from tests.helpers import replay_game_entry, synthetic_evals
from tests.test_gpu_engine import _oracle_follow, compare_all

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_arena_lockstep(games, visits, max_plies, iterations, seed):
    ocfg = orc.make_config(games=games, visits=visits, seed=seed, fen_str=orc.START_FEN_PLAIN, max_plies=max_plies,
                           weight=0.0, flags=orc.FLAG_ARENA)
    gcfg = link.Config(**{n: getattr(ocfg, n) for n, _ in orc.Config._fields_})
    oe, ge = orc.Engine(ocfg), link.Engine(gcfg)
    o_games, g_lines = [], []
    classes = set()
    for it in range(iterations):
        n_o, need_o = oe.select()
        n_g = ge.select()
        need_g, lb_g = ge.leaves()
        assert n_o == n_g and (need_o == need_g).all(), it
        classes |= set(need_o.tolist())
        logits, values = synthetic_evals(oe.leaf_boards())
        oe.backup(logits, values)
        ge.set_evals(logits, values)
        ge.backup()
        if it % 197 == 0:
            compare_all(oe, ge, range(games))
        o_games += oe.pop_games()
        g_lines += ge.drain_json()
    compare_all(oe, ge, range(games))
    assert {1, 2} <= classes <= {0, 1, 2}
    so, sg = oe.stats(), ge.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert so["reroot_nodes"] == 0  # fresh tree every ply
    assert len(g_lines) == len(o_games)
    parsed = sorted((json.loads(l) for l in g_lines), key=lambda e: e["uid"])
    for e, rec in zip(parsed, sorted(o_games, key=lambda r: r["uid"])):
        assert e["slot"] == rec["slot"] and e["uid"] == rec["uid"]
        assert {k: e[k] for k in ("boards", "dists", "moves", "result")} == rec["entry"]
    return [json.loads(l) for l in g_lines], so


def test_arena_search_matches_oracle_bit_for_bit():
    games, stats = run_arena_lockstep(games=10, visits=8, max_plies=400, iterations=4200, seed=31)
    assert len(games) >= 8 and stats["dropped"] == 0
    for e in games:
        assert e["result"] in (1, 2)
        assert replay_game_entry({"boards": e["boards"], "moves": e["moves"]}, orc.START_FEN_PLAIN) == e["result"]


def test_arena_games_cut_at_max_plies_are_reported_as_annulled():
    games, stats = run_arena_lockstep(games=6, visits=6, max_plies=30, iterations=700, seed=4)
    assert len(games) >= 6 and stats["dropped"] == len([e for e in games if e["result"] == 0]) > 0
    for e in games:
        if e["result"] == 0:
            assert len(e["moves"]) == 30


def test_batched_two_net_match_bookkeeping():
    conv, bn = model.random_init(2, 128, seed=21)
    conv2, bn2 = model.random_init(2, 128, seed=22)
    m = arena.Match((conv, bn), (conv2, bn2), visits=16, games=128, dtype="bf16", seed=5, max_plies=300)
    games = []
    for _ in range(400):
        m.run(50)
        games += m.drain()
        if len(games) >= 100:
            break
    st = m.engine.stats()
    m.close()
    assert len(games) >= 100 and st["reroot_nodes"] == 0
    assert {g["white"] for g in games} == {"a", "b"}  # every pairing is played both ways
    finished = [g for g in games if g["result"] in (1, 2)]
    assert len(finished) >= 3 and all(g["result"] in (0, 1, 2) for g in games)
    for g in finished[:30]:
        assert replay_game_entry({"boards": g["boards"], "moves": g["moves"]}, orc.START_FEN_PLAIN) == g["result"]
        x, o = g["final_score"]
        assert 0 <= x <= 49 and 0 <= o <= 49 and (x + o == 49 or x == 0 or o == 0 or g["result"] in (1, 2))
    for g in games:
        if g["result"] == 0:
            assert len(g["moves"]) == 300  # cut -> annulled


@pytest.mark.parametrize("dtype,blocks_b,games,thin", [
    ("f16", 2, 202, -1),    # at most 512 slots: the engine picks one board per workgroup — k_tower2_pair<Geo2Thin>
    ("bf16", 3, 202, 0),    # the same engine made to use the 3-board workgroups
    ("f16", 3, 602, -1),    # more than 512 slots: the 3-board k_tower2_pair, as the first ~490 games of a config-5 match run
    ("bf16", 2, 602, 1),    # ... and that engine switched to thin batches, as arena.Match does for a match's last games
])
def test_two_nets_in_one_tower_launch_equal_two_launches(dtype, blocks_b, games, thin, monkeypatch):
    """The arena's evaluator (uai_ringmaster.py:221-265: each position goes to the net whose move it is): both nets' leaf
    lists in ONE launch of the fused tower (k_tower2_pair; workgroups pick their weight set from the list they serve)
    against the two launches back to back it replaces (AZH_ARENA_PAIR=0) — same games, same trees, every word: a board's
    result does not depend on which launch carried it.  Nets of different depth included (blocks is per workgroup); both
    geometries of the kernel (3 boards per workgroup, one board per workgroup), each chosen by the engine's size and by hand."""
    wa = model.random_init(2, 128, seed=21)
    wb = model.random_init(blocks_b, 128, seed=22)
    sample = range(0, games, 7)
    runs = []
    for pair in ("1", "0"):
        monkeypatch.setenv("AZH_ARENA_PAIR", pair)
        m = arena.Match(wa, wb, visits=12, games=games, dtype=dtype, seed=9, max_plies=80)
        m.engine.set_thin_batches(thin)
        lines = []
        for _ in range(8):
            m.run(120)
            lines += m.engine.drain_json()
        m.engine.sync()
        states = [m.engine.game_state(g).as_tuple() for g in range(games)]
        trees = [m.engine.tree(g) for g in sample]
        runs.append((lines, states, trees, m.engine.stats()))
        m.close()
    (l1, s1, t1, st1), (l0, s0, t0, st0) = runs
    assert st1 == st0 and st1["plies"] > games * 4 and st1["games"] + st1["dropped"] > games // 4
    assert l1 == l0 and s1 == s0
    for a, b in zip(t1, t0):
        for x, y in zip(a, b):
            assert (x == y).all()


def _arena_oracle(m, games, visits, seed, max_plies=400):
    """The oracle engine of an arena.Match: same config, same loaded openings, same game limit."""
    ocfg = orc.make_config(games=games, visits=visits, seed=seed, fen_str=orc.START_FEN_PLAIN, max_plies=max_plies,
                           weight=0.0, flags=orc.FLAG_ARENA)
    for n, _ in orc.Config._fields_:
        assert getattr(ocfg, n) == getattr(m.engine.cfg, n), n
    return orc.Engine(ocfg)


def _match_in_lock_step(m, oe, net_a, net_b, dtype, thin, rounds, round_iters, sync_rounds, follow_thin_after_switch=False):
    """arena.Match's own host loop — fetch, enqueue the next round, then parse and score what was fetched, the thin switch
    decided at a drain (uai_ringmaster.py's drop-in and bench.py's config5 leg run exactly this) — with the oracle following
    iteration for iteration: every line of every round, and every state and arena word at the rounds in `sync_rounds`.
    -> (lines, the round after whose drain the match switched to thin batches or None)."""
    blockers = oe.cfg.blockers
    lines, switched = [], None
    followed = 0
    thin_from = None                                # first chunk of iterations that was enqueued after the switch
    m.run(round_iters)                              # chunk 0; round r enqueues chunk r + 1 and follows chunk r
    enqueued = round_iters

    def follow(chunk):
        nonlocal followed
        thin_now = thin or (follow_thin_after_switch and thin_from is not None and chunk >= thin_from)
        _oracle_follow(oe, net_a, blockers, enqueued - followed, dtype=dtype, thin=thin_now, net_b=net_b)
        followed = enqueued

    for r in range(rounds):
        m.fetch()                                   # waits for the iterations enqueued so far (chunks 0 .. r)
        if r in sync_rounds:                        # (nothing in flight here: states and arenas can be read)
            follow(r)
            compare_all(oe, m.engine, range(oe.G))
        m.run(round_iters)                          # chunk r + 1 runs while the finished games are parsed and scored
        if followed < enqueued:
            follow(r)
        enqueued += round_iters
        o_chunk = sorted(oe.pop_games(partial=True), key=lambda g: g["uid"])
        was_thin = m.thin
        g_chunk = m.drain()
        if m.thin and not was_thin:
            switched = r
            thin_from = r + 2                       # (chunk r + 1 was enqueued before this drain: still the 3-board kernel)
        assert len(g_chunk) == len(o_chunk), r
        for g, rec in zip(g_chunk, o_chunk):
            opening = g["opening"]
            assert g["uid"] == rec["uid"] and g["moves"][len(opening):] == rec["entry"]["moves"], r
            assert g["boards"] == rec["entry"]["boards"] and g["result"] == rec["entry"]["result"], r
        lines += g_chunk
    # the chunk still in flight: the oracle follows it, and its games are the last ones
    follow(rounds)
    m.fetch()
    o_chunk = sorted(oe.pop_games(partial=True), key=lambda g: g["uid"])
    g_chunk = m.drain()
    assert [(g["uid"], g["moves"][len(g["opening"]):], g["result"]) for g in g_chunk] == \
           [(rec["uid"], rec["entry"]["moves"], rec["entry"]["result"]) for rec in o_chunk]
    return lines + g_chunk, switched


@pytest.mark.parametrize("dtype", ["f16"])   # ("f32" passes too — 38 s: one launch per net; that path is held by the pair == two-launches test)
def test_arena_device_loop_matches_oracle_at_config5_size(dtype):
    """BASELINE configs[4] as bench.py's config5 leg and uai_ringmaster.py's drop-in run it — azh_engine_run_arena: 1000 games
    in flight, 100 visits per move from a fresh tree, two 12x128 nets, one leaf list per net, a fixed cohort under the game
    limit (slots go idle), arena.Match's fetch -> run -> drain order and its switch to thin batches once at most 512 games
    are left — against the oracle's arena mode (engine.py:474-530, uai_ringmaster.py:221-265) whose leaves go to net A or
    net B by the side to move: every game of every round, every state and arena word at four sync points, one of them after
    the switch.  f32: one launch per net, bit-identical wherever a board sits.  f16 — the leg's own dtype: both nets' lists in
    ONE launch of k_tower2_pair, three boards per workgroup until the switch, one board per workgroup after it; the oracle's
    leaves go through azh_net_forward (before the switch) / azh_net_forward_thin (after it) list by list in game order, which
    puts every board in the slot it has in the pair launch (each list starts at a workgroup of its own).  The match starts from
    random openings of 170 plies (uai_ringmaster.get_opening) so that games end, slots idle and the batch thins within a
    few thousand iterations instead of forty thousand."""
    G, V, seed = 1000, 100, 20260101
    wa, wb = model.random_init(12, 128, seed=1), model.random_init(12, 128, seed=2)
    m = arena.Match(wa, wb, V, games=G, dtype=dtype, seed=seed, opening_depth=170)
    oe = _arena_oracle(m, G, V, seed)
    oe.set_positions(np.repeat(m.opening_boards, 2, axis=0), np.full(G, 170, dtype=np.int32))
    m.set_game_limit(G)
    oe.set_game_limit(G)
    rounds = 44
    lines, switched = _match_in_lock_step(m, oe, m.net_a, m.net_b, link.DTYPES[dtype], False, rounds, 50,
                                          sync_rounds=(1, 14, 29, rounds - 1), follow_thin_after_switch=True)
    compare_all(oe, m.engine, range(G))
    so, sg = oe.stats(), m.engine.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    idle = sum(m.engine.game_state(g).phase == 3 for g in range(G))
    assert switched is not None and switched < rounds - 1, (switched, len(lines))   # a sync point lies behind the switch
    assert len(lines) >= G - link.THIN_MAX_GAMES and idle == len(lines) and sg["ring_overflow"] == 0
    assert {g["white"] for g in lines} == {"a", "b"} and len({g["result"] for g in lines}) >= 2
    m.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_arena_thin_pair_kernel_matches_oracle_bit_for_bit(dtype):
    """The arena's last games as they really run: both nets' leaf lists in one launch of the one-board-per-workgroup tower
    (k_tower2_pair<Geo2Thin>), 16-bit — pinned to the oracle with azh_net_forward_thin as its evaluator for either net (the
    thin kernel's result for a board does not depend on where it sits or which launch carries it).  A match that starts
    above 512 games is followed through its switch: 3-board kernel results cannot be reproduced board by board from the
    host (a 16-bit board's last bits depend on its slot in the workgroup), so the followed part begins at the switch."""
    G, V, seed = 384, 24, 77
    wa, wb = model.random_init(3, 128, seed=31), model.random_init(2, 128, seed=32)
    m = arena.Match(wa, wb, V, games=G, dtype=dtype, seed=seed, opening_depth=150)
    oe = _arena_oracle(m, G, V, seed)
    oe.set_positions(np.repeat(m.opening_boards, 2, axis=0), np.full(G, 150, dtype=np.int32))
    m.set_game_limit(G)
    oe.set_game_limit(G)
    lines, _ = _match_in_lock_step(m, oe, m.net_a, m.net_b, link.DTYPES[dtype], True, 30, 50, sync_rounds=(0, 9, 29))
    so, sg = oe.stats(), m.engine.stats()
    for k in so:
        assert so[k] == sg[k], (k, so[k], sg[k])
    assert len(lines) > 30 and sg["nn_evals"] > 20 * G and sg["ring_overflow"] == 0
    m.close()


def test_match_from_random_openings(tmp_path):
    """uai_ringmaster.get_opening (uai_ringmaster.py:185-196, OPENING_DEPTH random plies, the same opening for both games of a
    pairing) as arena.Match(opening_depth=N) / `uai_ringmaster.py --opening-depth N`: the slots are loaded with the positions
    after the openings; the games come back with the opening in front of their moves, replay from the start position, and
    the two games of a pairing share their opening while different pairings do not."""
    nets = [model.random_init(2, 128, seed=s) for s in (21, 22)]
    m = arena.Match(nets[0], nets[1], visits=16, games=64, dtype="f16", seed=9, max_plies=400, opening_depth=4)
    done = {}
    for _ in range(600):
        m.run(50)
        for g in m.drain():
            if g["uid"] < 64:
                done[g["uid"]] = g
        if len(done) == 64:
            break
    m.close()
    assert len(done) == 64
    for uid, g in done.items():
        assert len(g["opening"]) == 4 and g["moves"][:4] == g["opening"] and g["opening"] == done[uid ^ 1]["opening"]
        assert g["white"] == ("a" if uid % 2 == 0 else "b")
        p = orc.pos_from_fen(orc.START_FEN_PLAIN)                  # the whole game is legal from the start position
        for mv in g["moves"]:
            assert orc.result(p) == 0 and mv in [orc.move_string(x) for x in orc.movegen(p)], (uid, mv)
            c = orc.move_from_string(mv)
            orc.lib().orc_makemove(p, c & 0xFF, c >> 8)
        assert orc.result(p) == g["result"]
        assert g["result"] != 0 or len(g["moves"]) == 400          # (a game cut at the ply limit is annulled, opening plies included)
        x, o = g["final_score"]
        cells = [int(v) for v in orc.board_cells(p)]
        assert (cells.count(1), cells.count(2)) == (x, o)
    assert len({tuple(g["opening"]) for g in done.values()}) > 20
    # the command line, with the reference's `with opening: [...]` line and [Opening "..."] tag filled in
    pa, pb = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    model.save_model(pa, *nets[0])
    model.save_model(pb, *nets[1])
    pgn = str(tmp_path / "out.pgn")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "uai_ringmaster.py"),
                          "--engine", "python uai_interface.py --network-path %s --visits 8" % pa,
                          "--engine", "python uai_interface.py --network-path %s --visits 8" % pb,
                          "--pgn-out", pgn, "--game-count", "20", "--opening-depth", "3"], cwd=ROOT, capture_output=True, timeout=400)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    openings = re.findall(r"with opening: \[([a-g1-7, ]+)\]", res.stdout.decode())
    assert len(openings) == 20 and all(len(o.split(", ")) == 3 for o in openings)
    text = open(pgn).read()
    assert len(re.findall(r'\[Opening "[a-g1-7, ]+"\]', text)) == 20
    res = subprocess.run([sys.executable, os.path.join(ROOT, "uai_ringmaster.py"),
                          "--engine", "python uai_interface.py --network-path %s --visits 8" % pa,
                          "--engine", "python uai_interface.py --network-path %s --visits 8" % pb,
                          "--game-count", "5000", "--opening-depth", "3"], cwd=ROOT, capture_output=True, timeout=400)
    assert res.returncode != 0 and b"--opening-depth needs the whole match in flight" in res.stderr


def test_uai_ringmaster_cli(tmp_path):
    conv, bn = model.random_init(1, 128, seed=2)
    conv2, bn2 = model.random_init(1, 128, seed=3)
    pa, pb = str(tmp_path / "model-001.npy"), str(tmp_path / "model-002.npy")
    model.save_model(pa, conv, bn)
    model.save_model(pb, conv2, bn2)
    pgn = str(tmp_path / "out.pgn")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "uai_ringmaster.py"),
                          "--engine", "python uai_interface.py --network-path %s --visits 16" % pa,
                          "--engine", "python uai_interface.py --network-path %s --visits 16" % pb,
                          "--pgn-out", pgn, "--game-count", "40", "--concurrent", "64"],
                         cwd=ROOT, capture_output=True, timeout=400)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    out = res.stdout.decode()
    wins = re.findall(r"Wins: ([0-9.]+) - ([0-9.]+) \(annulled: (\d+)\)", out)
    assert len(wins) == 40
    a, b, ann = float(wins[-1][0]), float(wins[-1][1]), int(wins[-1][2])
    assert a + b == 40 and ann <= 40
    text = open(pgn).read()
    assert text.count('[Event "?"]') == 40 and text.count("[FinalScore ") == 40
    assert re.search(r'\[Result "(1-0|0-1|1/2-1/2)"\]', text)
    assert 'model-001.npy' in text and 'model-002.npy' in text


def test_config5_size_match_is_deterministic():
    """BASELINE configs[4]: a 1000-game arena between two 12x128 nets with deterministic seeds — played twice, the fixed
    cohort (uids 0..999, every pairing both ways) must come out identical game for game, and scoring a fixed cohort
    instead of the first finishers counts every game once."""
    nets = [model.random_init(12, 128, seed=s) for s in (1, 2)]

    def play():
        m = arena.Match(nets[0], nets[1], visits=24, games=1000, dtype="bf16", seed=20260101, max_plies=120)
        done = {}
        for _ in range(4000):
            m.run(50)
            for g in m.drain():
                if g["uid"] < 1000:
                    done[g["uid"]] = (tuple(g["moves"]), g["result"], g["white"])
            if len(done) == 1000:
                break
        m.close()
        return done

    a, b = play(), play()
    assert len(a) == 1000 and a == b
    assert sum(1 for v in a.values() if v[2] == "a") == 500   # each net has x in half of the cohort
    assert len({v[0] for v in a.values()}) > 20                # (no root noise in the arena: many pairings repeat a line)
