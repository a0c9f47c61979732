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


# strides coprime to 833 = 7 * 7 * 17: k -> (k * stride + offset) % 833 is a permutation of the policy indices
_STRIDES = np.array([s for s in range(2, 200) if s % 7 and s % 17], dtype=np.int64)


def synthetic_evals_distinct(leaf_boards):
    """Pure-function evaluator whose 833 logits are pairwise DISTINCT for every board (a board-dependent permutation
    of an arithmetic progression over [-3, 3)), so PUCT never meets two equal priors and the search is independent of
    any tie rule.  This is the evaluator injected into the reference's engine.py by tests/golden/gen_engine_fixtures.py
    and fed to the oracle / the HIP engine by the tests that replay those fixtures."""
    lb = np.asarray(leaf_boards, dtype=np.uint64).reshape(-1, 2)
    a = (lb[:, 0] % np.uint64(1000003)).astype(np.int64)
    b = (lb[:, 1] % np.uint64(999983)).astype(np.int64)
    off = (a * 31 + b * 17) % 833
    stride = _STRIDES[(a + 3 * b) % len(_STRIDES)]
    k = np.arange(833, dtype=np.int64)
    p = (k[None, :] * stride[:, None] + off[:, None]) % 833
    logits = p.astype(np.float32) * np.float32(6.0 / 833.0) - np.float32(3.0)
    values = (((a * 31 + b * 17) % 2001).astype(np.float32) / np.float32(1000.0)) - np.float32(1.0)
    return np.ascontiguousarray(logits, dtype=np.float32), np.ascontiguousarray(values, dtype=np.float32)


def leaf_boards_from_features(features):
    """(n,7,7,4) reference feature rows (engine.py:53-73) -> (n,2) u64 (mover, opponent), square = x + 7 * (6 - y)."""
    f = np.asarray(features).reshape(-1, 7, 7, 4)
    out = np.zeros((len(f), 2), dtype=np.uint64)
    for x in range(7):
        for y in range(7):
            sq = np.uint64(x + 7 * (6 - y))
            out[:, 0] |= (f[:, x, y, 1] != 0).astype(np.uint64) << sq
            out[:, 1] |= (f[:, x, y, 2] != 0).astype(np.uint64) << sq
    return out


def linear_evals(features):
    """Asymmetric pure-function evaluator on feature rows (used to pin the symmetry averaging of nn_evals.py:48-62):
    policy = features @ A, value = tanh(features @ b) with integer-built A, b (no RNG stream to drift)."""
    f = np.asarray(features, dtype=np.float64).reshape(-1, 196)
    i = np.arange(196, dtype=np.int64)[:, None]
    j = np.arange(833, dtype=np.int64)[None, :]
    A = (((i * 131 + j * 71 + i * j) % 257) - 128).astype(np.float64) / 128.0
    bvec = (((np.arange(196, dtype=np.int64) * 37) % 101) - 50).astype(np.float64) / 400.0
    return (f @ A).reshape(-1, 7, 7, 17), np.tanh(f @ bvec).reshape(-1, 1)
