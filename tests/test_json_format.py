"""The game line the host writes for a finished-game record (csrc/json.cpp) — checked on the CPU, no device needed.

The reference's client builds each entry as an nlohmann::json object and writes `entry.dump()` + newline
(cpp/self_play_client.cpp:512,565-578,637-642): keys sorted, no whitespace, floats as the shortest digits that round-trip,
laid out by nlohmann's format_buffer (plain decimals between 1e-4 and 1e15, d.ddde-XX outside).  `json.dumps(entry,
sort_keys=True, separators=(",", ":"))` writes the same bytes for values in [0, 1] — Python's repr(float) uses the same
digits and the same lay-out there — so it is the independent statement these lines are compared with, byte for byte.
"""
import ctypes
import json

import numpy as np
import pytest

from ataxxzero_amd import link

MAGIC = 0x415A4847


def cell_name(c):
    return "abcdefg"[c % 7] + "1234567"[c // 7]


def move_name(frm, to):
    return cell_name(to) if frm == to else cell_name(frm) + cell_name(to)


def random_record(rng, plies, visits_hi, random_ply=None, slot=3, uid=77, result=1, zero_total_at=None):
    """(record words as the device loop leaves them in its ring, the entry the reference would have dumped)"""
    words = [MAGIC, slot, uid, plies, result, 0, 0 if random_ply is None else random_ply + 1, 0]
    entry = {"boards": [], "dists": [], "moves": [], "result": result}
    if random_ply is not None:
        entry["random_ply"] = random_ply
    for p in range(plies):
        cells = rng.integers(0, 3, size=49)                    # 0 empty, 1 x, 2 o; bit = file + 7 * rank
        x = sum(1 << i for i in range(49) if cells[i] == 1)
        o = sum(1 << i for i in range(49) if cells[i] == 2)
        entry["boards"].append([int(cells[xx + 7 * (6 - y)]) for y in range(7) for xx in range(7)])
        nd = int(rng.integers(0, 40))
        pairs = set()
        while len(pairs) < nd:
            pairs.add((int(rng.integers(0, 49)), int(rng.integers(0, 49))))
        pairs = sorted(pairs)
        rng.shuffle(pairs)
        visits = [int(v) for v in rng.integers(0, visits_hi + 1, size=nd)]
        if zero_total_at == p:
            visits = [0] * nd
        frm, to = (int(rng.integers(0, 49)), int(rng.integers(0, 49))) if not pairs else pairs[0]
        words += [x & 0xFFFFFFFF, x >> 32, o & 0xFFFFFFFF, o >> 32, (frm | to << 8) | nd << 16, 0]
        words += [(f | t << 8) | v << 16 for (f, t), v in zip(pairs, visits)]
        total = sum(visits)
        entry["dists"].append({move_name(f, t): (v / total if total else 0.0) for (f, t), v in zip(pairs, visits)})
        entry["moves"].append(move_name(frm, to))
    words[5] = len(words)
    return np.array(words, dtype=np.uint32), entry


def dumped(entry):
    return json.dumps(entry, sort_keys=True, separators=(",", ":")).encode()


@pytest.mark.parametrize("visits_hi", [1, 7, 400, 2000, 60000])
def test_lines_equal_the_sorted_compact_dump_byte_for_byte(visits_hi):
    rng = np.random.default_rng(visits_hi)
    for case in range(40):
        rec, entry = random_record(rng, plies=int(rng.integers(1, 30)), visits_hi=visits_hi,
                                   random_ply=int(rng.integers(0, 120)) if case % 4 == 3 else None,
                                   result=1 + case % 2, zero_total_at=0 if case % 10 == 9 else None)
        assert link.format_record_json(rec) == dumped(entry)
        with_ids = dict(entry, slot=3, uid=77)
        assert link.format_record_json(rec, with_ids=True) == dumped(with_ids)


def test_small_ratios_are_written_as_nlohmann_writes_them():
    """1 / 2000 is "0.0005" (not the shorter "5e-04"), 1 / 10000 "0.0001", 1 / 40000 "2.5e-05", 1 / 60000 the
    seventeen-digit form: the boundary of the plain notation is 1e-4, and exponents carry two digits."""
    for total, want in ((2000, b"0.0005"), (10000, b"0.0001"), (40000, b"2.5e-05"), (60000, b"1.6666666666666667e-05"),
                        (3, b"0.3333333333333333"), (1, b"1.0"), (400, b"0.0025")):
        frm, to = 0, 8
        head = [MAGIC, 0, 0, 1, 1, 0, 0, 0]
        ply = [1, 0, 2, 0, (frm | to << 8) | 2 << 16, 0, (frm | to << 8) | 1 << 16, (1 | 9 << 8) | (total - 1) << 16]
        if total == 1:
            ply[4] = (frm | to << 8) | 1 << 16
            ply = ply[:7]
        rec = np.array(head + ply, dtype=np.uint32)
        rec[5] = len(rec)
        line = link.format_record_json(rec)
        assert b'"a1b2":' + want + (b"," if total > 1 else b"}") in line, (total, line)
        assert json.loads(line)["dists"][0]["a1b2"] == 1 / total


def test_malformed_records_are_refused():
    rng = np.random.default_rng(5)
    rec, _ = random_record(rng, plies=4, visits_hi=50)
    bad = rec.copy()
    bad[0] = 0
    with pytest.raises(link.AzhError):
        link.format_record_json(bad)
    with pytest.raises(link.AzhError):
        link.format_record_json(rec[:-1])          # the record says it is longer than what was handed over
    cut = rec.copy()
    cut[3] += 1                                    # one ply more than the words hold
    with pytest.raises(link.AzhError):
        link.format_record_json(cut)
    # a buffer that is too small reports the size it needs instead of writing a partial line
    need = ctypes.c_int64(0)
    buf = np.zeros(16, dtype=np.uint8)
    rc = link.load().azh_format_record_json(rec.ctypes.data, len(rec), 0, buf.ctypes.data, buf.nbytes, ctypes.byref(need))
    assert rc == -6 and need.value == len(link.format_record_json(rec))
