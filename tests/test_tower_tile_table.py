"""The tower's cell-tile table in csrc/net_kernels.hip is the one tools/tower_tile_table.py generates, and it has the
properties the kernel's compile-time tap skipping relies on (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(text):
    body = text[text.index("TILE_CELL[10][16] = {"):]
    body = body[:body.index("};")]
    return [[int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", line)] for line in body.splitlines()[1:] if "0x" in line]


def test_source_table_is_the_generated_one():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tower_tile_table.py")], capture_output=True, text=True,
                         check=True).stdout
    with open(os.path.join(ROOT, "ataxxzero_amd", "csrc", "net_kernels.hip")) as f:
        src = f.read()
    assert _rows(out) == _rows(src)


def test_table_properties():
    with open(os.path.join(ROOT, "ataxxzero_amd", "csrc", "net_kernels.hip")) as f:
        rows = _rows(f.read())
    assert len(rows) == 10 and all(len(r) == 16 for r in rows)
    cells = [c for r in rows for c in r if c < 0x100]
    assert sorted(cells) == list(range(147))                      # every cell of three boards exactly once
    for r in rows:
        for lane, c in enumerate(r):
            assert (c & 0xFF) % 16 == lane                        # bank rule: lane r holds a slot with slot % 16 == r

    def xy(c):                                                    # c = 21 y + 7 board + x
        return (c % 21) % 7, c // 21

    # wave half 0 (tiles 0-4, inner tap index = dy): tile 0 on y = 0, tiles 1 and 4 on y = 6
    assert all(xy(c)[1] == 0 for c in rows[0] if c < 0x100)
    assert all(xy(c)[1] == 6 for c in rows[1] if c < 0x100)
    assert all(xy(c)[1] == 6 for c in rows[4] if c < 0x100) and sum(c < 0x100 for c in rows[4]) == 3
    # wave half 1 (tiles 5-9, inner tap index = dx): tile 5 on x = 0, tile 6 on x = 6
    assert all(xy(c)[0] == 0 for c in rows[5] if c < 0x100)
    assert all(xy(c)[0] == 6 for c in rows[6] if c < 0x100)


def test_thin_geometry_table_covers_board_zero_once_and_keeps_the_bank_rule():
    """TILE_CELL1 (Geo2Thin: one board per workgroup): the 49 cells of board 0 — slots 21 y + x of the same LDS image — exactly
    once over four 16-lane tiles, lane r again a slot with slot % 16 == r, empty lanes marked 0x100 | r."""
    with open(os.path.join(ROOT, "ataxxzero_amd", "csrc", "net_kernels.hip")) as f:
        text = f.read()
    body = text[text.index("TILE_CELL1[4][16] = {"):]
    body = body[:body.index("};")]
    rows = [[int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", line)] for line in body.splitlines()[1:] if "0x" in line]
    assert len(rows) == 4 and all(len(r) == 16 for r in rows)
    cells = sorted(c for r in rows for c in r if c < 0x100)
    assert cells == sorted(21 * y + x for y in range(7) for x in range(7))
    for r in rows:
        for lane, c in enumerate(r):
            assert (c & 0xFF) % 16 == lane
            assert c < 0x100 or c == 0x100 | lane
