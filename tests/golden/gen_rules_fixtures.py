#!/usr/bin/env python3
"""Generate the golden rule vectors from the reference's own Python rules.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONHASHSEED=0 python tests/golden/gen_rules_fixtures.py
    PYTHONHASHSEED=0 python tests/golden/gen_engine_fixtures.py     # engine_features.npz follows the position order

Reproducible since round 4: the reference's `legal_moves()` builds its clone moves from a `set` of ("c", dest) tuples
(ataxx_rules.py:134-157), so the ORDER of the list depends on the interpreter's string-hash seed; every draw this script
makes from such a list (the move played, the successors recorded) is therefore made from the list sorted by UAI text, and
the script refuses to run under any hash seed but 0, the one gen_engine_fixtures.py is run under.  (Rounds 1-3 drew from the
unsorted list under an unrecorded hash seed: the vectors were the reference's own answers, but the recipe did not give the
same positions twice.)

Imports `ataxx_rules` and `uai_interface` from /root/reference unmodified
(stdlib-only modules) and writes data-only fixtures next to this script:

  perft.json            perft node counts (perft.py:5-26 semantics) for the
                        no-blocker start and the 4-blocker self-play start
  rules_noblock.json.gz positions from seeded random games, no blockers
  rules_block4.json.gz  same with the self-play blockers (module constant
                        ataxx_rules.BLOCKED_CELLS set to the 4 cells the
                        reference keeps commented out at ataxx_rules.py:9)
  policy_layers.json    (dx,dy) -> policy layer, from the enumeration order of
                        ataxx_rules.FAR_NEIGHBOR_OFFSETS (engine.py:75)
  uai_codec.json        uai_interface encode/decode examples

Each position record: fen, to_move, sorted UAI move list (or ["0000"] when the
side must pass), result (0 ongoing / 1 / 2), the 49-cell board list, and for a
sample of moves the successor FEN.
"""
import gzip
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference")
import ataxx_rules  # noqa: E402
import uai_interface  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK4 = frozenset([(3, 2), (2, 3), (4, 3), (3, 4)])


def set_blockers(cells):
    ataxx_rules.BLOCKED_CELLS = frozenset(cells)
    ataxx_rules.LEGAL_SQUARE_COUNT = ataxx_rules.SIZE * ataxx_rules.SIZE - len(cells)


def perft(position, depth):
    # perft.py:5-16, without the prints.
    ensemble = [position.copy()]
    for _ in range(depth):
        new_ensemble = []
        for b in ensemble:
            for move in b.legal_moves():
                nb = b.copy()
                nb.move(move)
                new_ensemble.append(nb)
        ensemble = new_ensemble
    return len(ensemble)


def perft_table(max_depth):
    board = ataxx_rules.AtaxxState.initial()
    table = {"fen": board.fen(), "depth": {}}
    for d in range(1, max_depth + 1):
        table["depth"][str(d)] = perft(board, d)
    # perft.py:18-26: per-root-move split, depth 1 + 3.
    split = {}
    for move in board.legal_moves():
        c = board.copy()
        c.move(move)
        split[uai_interface.uai_encode_move(move)] = perft(c, 3)
    table["split_d4"] = split
    return table


def in_uai_order(moves):
    """the list `legal_moves()` returns, in an order that does not depend on the hash seed"""
    return sorted(moves, key=uai_interface.uai_encode_move)


def position_record(state, rng):
    moves = state.legal_moves()
    rec = {
        "fen": state.fen(),
        "to_move": state.to_move,
        "moves": sorted(uai_interface.uai_encode_move(m) for m in moves),
        "result": state.result() or 0,
        "cells": list(state.board),
    }
    succ = {}
    for m in rng.sample(in_uai_order(moves), min(4, len(moves))):
        c = state.copy()
        c.move(m)
        succ[uai_interface.uai_encode_move(m)] = c.fen()
    rec["succ"] = succ
    return rec


def random_positions(seed, n_games, keep_prob):
    rng = random.Random(seed)
    out = []
    for _ in range(n_games):
        state = ataxx_rules.AtaxxState.initial()
        for _ply in range(400):
            if rng.random() < keep_prob:
                out.append(position_record(state, rng))
            state.move(rng.choice(in_uai_order(state.legal_moves())))
            if state.result() is not None:
                out.append(position_record(state, rng))  # terminal position
                break
    return out


def dump_gz(name, obj):
    # mtime=0 keeps the archive byte-stable across regenerations.
    with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as f:
        f.write(json.dumps(obj, separators=(",", ":")).encode())


def main():
    if os.environ.get("PYTHONHASHSEED") != "0":
        raise SystemExit("run as: PYTHONHASHSEED=0 python tests/golden/gen_rules_fixtures.py (see the docstring)")
    perfts = {}
    set_blockers(frozenset())
    perfts["noblock"] = perft_table(5)
    dump_gz("rules_noblock.json.gz", random_positions(1234, 60, 0.2))
    layers = [[dx, dy, i] for i, (dx, dy) in enumerate(ataxx_rules.FAR_NEIGHBOR_OFFSETS)]

    set_blockers(BLOCK4)
    perfts["block4"] = perft_table(4)
    dump_gz("rules_block4.json.gz", random_positions(4321, 40, 0.2))
    set_blockers(frozenset())

    with open(os.path.join(HERE, "perft.json"), "w") as f:
        json.dump(perfts, f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "policy_layers.json"), "w") as f:
        json.dump(layers, f)

    codec = []
    rng = random.Random(7)
    for _ in range(64):
        a = (rng.randrange(7), rng.randrange(7))
        b = (rng.randrange(7), rng.randrange(7))
        for m in (("c", a), (a, b)):
            s = uai_interface.uai_encode_move(m)
            back = uai_interface.uai_decode_move(s)
            codec.append({"move": [m[0] if m[0] == "c" else list(m[0]), list(m[1])],
                          "uai": s, "decoded": [back[0] if back[0] == "c" else list(back[0]), list(back[1])]})
    codec.append({"move": "pass", "uai": uai_interface.uai_encode_move("pass"), "decoded": "pass"})
    with open(os.path.join(HERE, "uai_codec.json"), "w") as f:
        json.dump(codec, f)
    print("perft:", json.dumps(perfts["noblock"]["depth"]), json.dumps(perfts["block4"]["depth"]))


if __name__ == "__main__":
    main()
