"""Shared helpers for the parity tests (no reference imports: /root/reference does not
exist on the GPU box; only tests/golden fixtures and the oracle are used)."""
import gzip
import json
import os

import numpy as np

from oracle import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BLOCK4_MASK = sum(1 << (x + 7 * (6 - y)) for x, y in [(3, 2), (2, 3), (4, 3), (3, 4)])


def load_gz(name):
    with gzip.open(os.path.join(GOLDEN, name)) as f:
        return json.loads(f.read())


def fixture_positions(limit=None):
    """[(packed board (2,) u64, blockers, record)] from both golden position files."""
    out = []
    for name, blockers in (("rules_noblock.json.gz", 0), ("rules_block4.json.gz", BLOCK4_MASK)):
        for rec in load_gz(name)[:limit]:
            p = orc.pos_from_fen(rec["fen"])
            packed = np.array([int(p.pieces[0]) | (p.turn << 63), int(p.pieces[1])], dtype=np.uint64)
            out.append((packed, blockers, rec))
    return out


def synthetic_evals(leaf_boards):
    """Deterministic pseudo-evaluator: f32 logits (G,833) and values (G,) from the leaf
    boards alone, so the GPU engine and the oracle can be fed identical bits."""
    lb = np.asarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
    a = (lb[:, 0] % np.uint64(1000003)).astype(np.int64)
    b = (lb[:, 1] % np.uint64(999983)).astype(np.int64)
    k = np.arange(833, dtype=np.int64)
    h = (a[:, None] * (k[None, :] + 1) + b[:, None] * (k[None, :] + 7) + 13 * k[None, :] * k[None, :]) % 4001
    logits = (h.astype(np.float32) / np.float32(1000.0)) - np.float32(2.0)
    values = (((a * 31 + b * 17) % 2001).astype(np.float32) / np.float32(1000.0)) - np.float32(1.0)
    return np.ascontiguousarray(logits, dtype=np.float32), np.ascontiguousarray(values, dtype=np.float32)


def replay_game_entry(entry, start_fen, blockers_mask=None):
    """Check a JSON game entry against the oracle rules; returns the final result."""
    p = orc.pos_from_fen(start_fen)
    if blockers_mask is not None:
        p.blockers = blockers_mask
    assert len(entry["boards"]) == len(entry["moves"])
    for i, (b, m) in enumerate(zip(entry["boards"], entry["moves"])):
        assert [int(v) for v in orc.board_cells(p)] == b, "board mismatch at ply %d" % i
        assert orc.result(p) == 0
        legal = [orc.move_string(x) for x in orc.movegen(p)]
        assert m in legal, (m, legal)
        if "dists" in entry:
            d = entry["dists"][i]
            # the ONE_RANDOM_MOVE ply may play a legal move the search never expanded
            assert m in d or entry.get("random_ply") == i
            assert abs(sum(d.values()) - 1.0) < 1e-9 and set(d) <= set(legal)
        c = orc.move_from_string(m)
        orc.lib().orc_makemove(p, c & 0xFF, c >> 8)
    return orc.result(p)
