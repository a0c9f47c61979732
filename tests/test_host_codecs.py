"""The product's host-side helpers (ataxxzero_amd/uai.py, selfplay.py, arena.py) against the vectors generated from the
reference's own Python — on the CPU: none of these functions touches the device, and the GPU tests only reach them through
whole CLI runs.

Vectors: tests/golden/uai_codec.json (uai_interface.py:11-32 encode/decode on ataxx_rules move tuples),
rules_noblock / rules_block4 (ataxx_rules.py positions: fen, cells, legal moves in UAI text, successors).
"""
import json
import os

import pytest

from ataxxzero_amd import arena, selfplay, uai
from helpers import load_gz


@pytest.fixture(scope="module")
def codec(golden_dir):
    with open(os.path.join(golden_dir, "uai_codec.json")) as f:
        return json.load(f)


def as_tuple(move):
    """json turned the reference's tuples into lists"""
    if move == "pass":
        return "pass"
    return tuple("c" if part == "c" else tuple(part) for part in move)


def test_uai_text_of_reference_moves(codec):
    """uai_interface.encode_move / decode_move on every move value the fixture holds"""
    assert len(codec) > 100
    for rec in codec:
        move = as_tuple(rec["move"])
        assert uai.xy_move_to_text(move) == rec["uai"]
        assert uai.text_to_xy_move(rec["uai"]) == as_tuple(rec["decoded"])
    assert uai.xy_move_to_text("pass") == "0000"
    for text in ("pass", "none", "0000"):
        assert uai.text_to_xy_move(text) == "pass"
    with pytest.raises(Exception):
        uai.text_to_xy_move("a1b")


def test_engine_move_codes_round_trip_through_uai_text(codec):
    """u16 from | to << 8 (the engine's move) <-> UAI text <-> the Python generator's move value (generate_games.py:51)"""
    for rec in codec:
        if rec["move"] == "pass" or rec["move"][0] == rec["move"][1]:
            continue    # (the fixture also holds the degenerate "jump" d4 -> d4, "d4d4": no legal move, and the engine's code
                        # for from == to is the clone)
        code = uai.decode_move(rec["uai"])
        assert uai.encode_move(code) == rec["uai"]
        # the jump/clone distinction of the reference's tuples is lost in UAI text for single steps (uai_interface.py
        # decodes "c6" as a clone to c6): compare through the decoded value the reference itself gives
        assert selfplay.python_move(code) == rec["decoded"]
        frm, to = code & 0xFF, code >> 8
        assert (frm == to) == (len(rec["uai"]) == 2)
    assert uai.decode_move("0000") == 0xFFFF and uai.encode_move(0xFFFF) == "0000"
    assert selfplay.python_move(0xFFFF) == "pass"
    assert uai.decode_move("A1") == uai.decode_move("a1")


@pytest.mark.parametrize("name", ["rules_noblock.json.gz", "rules_block4.json.gz"])
def test_fen_parser_and_cells_on_reference_positions(name):
    recs = load_gz(name)
    for rec in recs[::3]:
        x, o, bl, turn = selfplay.parse_fen(rec["fen"])
        assert x & o == 0 and x & bl == 0 and o & bl == 0
        assert turn == (0 if rec["to_move"] == 1 else 1)
        cells = selfplay.board_cells(x, o)
        want = [c if c in (1, 2) else 0 for c in rec["cells"]]    # game files write blockers as 0
        assert cells == want
        if bl == 0:
            assert uai.Position(x, o, turn).fen() == rec["fen"]
    assert selfplay.parse_fen(selfplay.START_FEN_SELFPLAY)[2] == (1 << 31) | (1 << 23) | (1 << 25) | (1 << 17)
    with pytest.raises(ValueError):
        selfplay.parse_fen("x5o/7/7/7/7/7/o5q x")


def test_final_score_replay_equals_the_reference_successor():
    """arena.replay_final_score applies the last move to the last board (the PGN's FinalScore tag,
    uai_ringmaster.py:162-180): checked against every successor ataxx_rules produced for x-to-move positions"""
    checked = 0
    for rec in load_gz("rules_noblock.json.gz"):
        if rec["to_move"] != 1 or not rec["succ"]:
            continue
        for move, fen_after in rec["succ"].items():
            if move in ("pass", "0000"):
                continue
            x, o, _, _ = selfplay.parse_fen(fen_after)
            entry = {"boards": [rec["cells"]], "moves": [move]}
            assert arena.replay_final_score(entry) == (bin(x).count("1"), bin(o).count("1")), (rec["fen"], move)
            checked += 1
    assert checked > 3000
    assert arena.replay_final_score({"boards": [], "moves": []}) == (2, 2)


def test_generator_file_index_follows_loopers_naming():
    """looper.py:70-74: games/model-%03i-%i.json — the trailing index picks the GPU and the RNG stream"""
    assert selfplay.process_index_from_path("games/model-007-3.json") == 3
    assert selfplay.process_index_from_path("/tmp/run/games/model-012-11.json") == 11
    assert selfplay.process_index_from_path("games/random-play.json") == 0
    assert selfplay.process_index_from_path("model-001-0.json") == 0
