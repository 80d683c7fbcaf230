"""What a TILE-ALIGNED ordering of the nested dissection would cost ("what a next round would have to change"):
today a front's boundary cells are packed densely (b cells + the right-hand-side row -> ceil((b + 1) / 16) tiles) and a child's update
matrix lands in its parent entry by entry (one LDS gather per entry and child, through host-built recipes).  If the boundary cells
were grouped by the ancestor separator they belong to and every group padded to whole tiles of 16, a child's update would land in
its parent as whole 16 x 16 tiles (tile-to-tile adds).  This script counts, from the symbolic tables (no device), the tiles and
matrix-core instructions per member of both orderings:
    python tests/tools/nd_tile_aligned_estimate.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd import _lib  # noqa: E402
from tests import nd_emulate as E  # noqa: E402

lib = _lib.load()
tab = E.tables(lib, 128, 128)
fronts, cells = tab["fronts"], tab["cells"]
nF = len(fronts)
owner = np.full(128 * 128, -1)  # front that eliminates the cell
for f in range(nF):
    F = fronts[f]
    cl = cells[F[E.F_CELLS]:F[E.F_CELLS] + 16 * (F[E.F_ST] + F[E.F_BT])]
    owner[cl[:F[E.F_S]]] = f


def mfma_count(st, bt):
    """v_mfma_f64_16x16x4 instructions of one front with st pivot tiles and bt boundary tiles (W = P V: 4 per tile; updates: 4 per
    tile pair and panel), as the kernels issue them (full 16-pivot panels)."""
    n = 0
    T = st + bt
    for pp in range(st):
        rows = T - pp - 1
        n += 4 * rows                      # W_R^T = P V_R
        n += 4 * rows * (rows + 1) // 2    # trailing tiles R >= C > pp
    return n


tot = {"dense": [0, 0, 0], "aligned": [0, 0, 0]}  # boundary tiles, mfma, update-matrix doubles
by_level = {}
for f in range(nF):
    F = fronts[f]
    lv, s, b, st, bt = (int(F[i]) for i in (E.F_LEVEL, E.F_S, E.F_B, E.F_ST, E.F_BT))
    if lv == 10:
        continue  # the leaves are eliminated in registers, one lane each: no tiles
    cl = cells[F[E.F_CELLS] + 16 * st:F[E.F_CELLS] + 16 * st + b]
    groups = {}
    for c in cl:
        groups[owner[c]] = groups.get(owner[c], 0) + 1
    # the right-hand-side row rides in the last group's padding if there is room, else in a tile of its own
    sizes = sorted(groups.values())
    bta = sum((g + 15) // 16 for g in sizes)
    if not any(g % 16 for g in sizes) or not sizes:
        bta += 1
    for key, btx in (("dense", bt), ("aligned", bta)):
        tot[key][0] += btx
        tot[key][1] += mfma_count(st, btx)
        tot[key][2] += (16 * btx) * (16 * btx + 1) // 2
    d = by_level.setdefault(lv, [0, 0, 0, 0])
    d[0] += 1; d[1] += bt; d[2] += bta; d[3] = max(d[3], len(sizes))
print("level  fronts  boundary tiles dense -> tile-aligned   (max ancestor groups per front)")
for lv in sorted(by_level):
    n, a, b_, g = by_level[lv]
    print(f"{lv:5d} {n:7d}  {a / n:6.2f} -> {b_ / n:6.2f}   ({g})")
for key in ("dense", "aligned"):
    bt_, mf, upd = tot[key]
    print(f"{key:8s}: {bt_} boundary tiles, {mf} matrix-core instructions per member ({mf * 512 / 1e6:.1f} Mflop), update matrices {upd * 8 / 1e6:.1f} MB")
