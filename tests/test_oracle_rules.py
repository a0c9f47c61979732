"""Pin the CPU oracle's rules to the vectors generated from the reference's own
Python rules (tests/golden/gen_rules_fixtures.py; ataxx_rules.py, perft.py)."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle_lib as orc


def load_gz(golden_dir, name):
    with gzip.open(os.path.join(golden_dir, name)) as f:
        return json.loads(f.read())


BLOCK4_MASK = sum(1 << (x + 7 * (6 - y)) for x, y in [(3, 2), (2, 3), (4, 3), (3, 4)])


def with_blockers(fen, blockers):
    p = orc.pos_from_fen(fen)
    p.blockers |= blockers
    return p


@pytest.mark.parametrize("name,blockers", [("rules_noblock.json.gz", 0), ("rules_block4.json.gz", BLOCK4_MASK)])
def test_positions_match_reference(golden_dir, name, blockers):
    recs = load_gz(golden_dir, name)
    assert len(recs) > 1000
    for rec in recs:
        p = with_blockers(rec["fen"], blockers)
        assert orc.fen(p) == rec["fen"]
        assert p.turn == rec["to_move"] - 1
        moves = orc.movegen(p)
        got = sorted(orc.move_string(m) for m in moves) or ["0000"]
        assert got == rec["moves"], rec["fen"]
        assert orc.result(p) == rec["result"], rec["fen"]
        # reference board list index = x + 7*y, y = 0 at rank 7 (ataxx_rules.py:74-80)
        assert [int(v) for v in orc.board_cells(p)] == rec["cells"]
        for mv, fen2 in rec["succ"].items():
            q = with_blockers(rec["fen"], blockers)
            if mv == "0000":
                orc.lib().orc_pass(q)
            else:
                code = orc.move_from_string(mv)
                orc.lib().orc_makemove(q, code & 0xFF, code >> 8)
            assert orc.fen(q) == fen2, (rec["fen"], mv)


def test_movegen_order_is_reference_order():
    # cpp/movegen.cpp:10-79: jumps ascending (from, to), then clones ascending.
    p = orc.pos_from_fen(orc.START_FEN_SELFPLAY)
    moves = orc.movegen(p)
    jumps = [(int(m) & 0xFF, int(m) >> 8) for m in moves if (int(m) & 0xFF) != (int(m) >> 8)]
    clones = [int(m) >> 8 for m in moves if (int(m) & 0xFF) == (int(m) >> 8)]
    assert jumps == sorted(jumps) and clones == sorted(clones)
    assert [((int(m) & 0xFF) != (int(m) >> 8)) for m in moves] == [True] * len(jumps) + [False] * len(clones)


def test_perft_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "perft.json")) as f:
        table = json.load(f)
    for key, blockers in (("noblock", 0), ("block4", BLOCK4_MASK)):
        t = table[key]
        p = with_blockers(t["fen"], blockers)
        for d, n in t["depth"].items():
            assert orc.perft(p, int(d)) == n, (key, d)
        for mv, n in t["split_d4"].items():
            q = with_blockers(t["fen"], blockers)
            code = orc.move_from_string(mv)
            orc.lib().orc_makemove(q, code & 0xFF, code >> 8)
            assert orc.perft(q, 3) == n
    # SURVEY.md §0 tables (C++ reference agreement at depth 6 is quoted there; d5 blockers here)
    assert table["noblock"]["depth"]["4"] == 155888
    assert orc.perft(with_blockers("x5o/7/7/7/7/7/o5x x", BLOCK4_MASK), 5) == 3639872


def test_masks_are_chebyshev_rings():
    for sq in range(49):
        f, r = sq % 7, sq // 7
        for d, fn in ((1, orc.lib().orc_singles), (2, orc.lib().orc_doubles)):
            want = 0
            for s2 in range(49):
                if max(abs(s2 % 7 - f), abs(s2 // 7 - r)) == d:
                    want |= 1 << s2
            assert fn(sq) == want
    # spot values quoted from cpp/bitboards.hpp:32-33 (a1, g7)
    assert orc.lib().orc_singles(0) == 0x182 and orc.lib().orc_doubles(0) == 0x1c204
    assert orc.lib().orc_singles(48) == 0x830000000000 and orc.lib().orc_doubles(48) == 0x408700000000


def test_policy_index_and_features(golden_dir):
    with open(os.path.join(golden_dir, "policy_layers.json")) as f:
        layers = {(dx, dy): i for dx, dy, i in json.load(f)}
    assert len(layers) == 16
    for frm in range(49):
        for to in range(49):
            fx, fy, tx, ty = frm % 7, 6 - frm // 7, to % 7, 6 - to // 7
            idx = orc.lib().orc_policy_index(frm | (to << 8))
            if frm == to:
                assert idx == 119 * tx + 17 * ty + 16
            elif max(abs(tx - fx), abs(ty - fy)) == 2:
                assert idx == 119 * tx + 17 * ty + layers[(tx - fx, ty - fy)]
    # known answers recorded in SURVEY.md §8(c) from engine.get_move_score's indexing
    assert orc.lib().orc_policy_index(orc.move_from_string("b7")) == 135
    assert orc.lib().orc_policy_index(orc.move_from_string("a7c6")) == 269
    assert orc.lib().orc_policy_index(orc.move_from_string("g1e2")) == 562
    f = orc.features(orc.pos_from_fen(orc.START_FEN_PLAIN))
    assert f.sum(axis=(0, 1)).tolist() == [49, 2, 2, 0]
    assert f[0, 0].tolist() == [1, 1, 0, 0]
    f = orc.features(orc.pos_from_fen(orc.START_FEN_SELFPLAY))
    assert f.sum(axis=(0, 1)).tolist() == [49, 2, 2, 4]
    p = orc.pos_from_fen("x5o/7/7/7/7/7/o5x o")
    assert orc.features(p)[6, 0].tolist() == [1, 1, 0, 0]  # g7 is o's stone, o to move


def test_uai_codec(golden_dir):
    with open(os.path.join(golden_dir, "uai_codec.json")) as f:
        recs = json.load(f)
    for rec in recs:
        if rec["move"] == "pass":
            continue
        start, end = rec["move"]
        to = end[0] + 7 * (6 - end[1])
        if start == "c":
            code = to | (to << 8)
            assert orc.move_string(code) == rec["uai"]
        elif max(abs(start[0] - end[0]), abs(start[1] - end[1])) == 2:
            frm = start[0] + 7 * (6 - start[1])
            assert orc.move_string(frm | (to << 8)) == rec["uai"]
