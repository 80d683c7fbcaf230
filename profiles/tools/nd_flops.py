#!/usr/bin/env python3
"""Flop and front-size census of a nested-dissection (multifrontal) factorisation of the n x n five-point pressure system, beside
the block elimination along ix that k_press128s performs -- the "fewer flops" candidate of DESIGN.md section 8, quantified.

    python3 profiles/tools/nd_flops.py [n=128] [leaf=8]

Geometric dissection: a region is cut across its longer side by a one-cell-wide separator until both sides are <= leaf; a region's
boundary is the part of its perimeter that is an ancestor's separator (domain borders carry none).  A front with s pivots and b
boundary unknowns costs  s^3/3 + s^2 b + s b^2  multiply-adds (Cholesky of the pivot block, the panel solve, the symmetric Schur
update counted in full as the matrix pipe would execute it); the substitution passes cost 2 (s^2 + 2 s b) per right-hand side."""
import sys
from collections import defaultdict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 8

fronts = []  # (level, s, b)


def dissect(w, h, sides, level):
    """sides = (west, east, south, north): True where the side borders an ancestor's separator."""
    b = (h if sides[0] else 0) + (h if sides[1] else 0) + (w if sides[2] else 0) + (w if sides[3] else 0)
    if w <= leaf and h <= leaf:
        fronts.append((level, w * h, b, "leaf"))
        return
    if w >= h:  # cut across x: separator of h cells
        wl = (w - 1) // 2
        wr = w - 1 - wl
        fronts.append((level, h, b, "sep"))
        dissect(wl, h, (sides[0], True, sides[2], sides[3]), level + 1)
        dissect(wr, h, (True, sides[1], sides[2], sides[3]), level + 1)
    else:
        hl = (h - 1) // 2
        hr = h - 1 - hl
        fronts.append((level, w, b, "sep"))
        dissect(w, hl, (sides[0], sides[1], sides[2], True), level + 1)
        dissect(w, hr, (sides[0], sides[1], True, sides[3]), level + 1)


dissect(n, n, (False, False, False, False), 0)
ma = lambda s, b: s ** 3 / 3 + s * s * b + s * b * b
tot = sum(ma(s, b) for _, s, b, _ in fronts)
solve = sum(2 * (s * s + 2 * s * b) for _, s, b, _ in fronts)
upd_mem = sum(b * b for _, s, b, _ in fronts) * 8
fac_mem = sum(s * s + s * b for _, s, b, _ in fronts) * 8
by_level = defaultdict(lambda: [0, 0, 0, 0.0])
for lv, s, b, kind in fronts:
    e = by_level[(lv, kind)]
    e[0] += 1
    e[1] = max(e[1], s)
    e[2] = max(e[2], s + b)
    e[3] += ma(s, b)
print(f"{n} x {n} grid, leaves <= {leaf} x {leaf}: {len(fronts)} fronts, {sum(s for _, s, _, _ in fronts)} unknowns")
print(" level kind  fronts  max pivots  max front   Mflop (2 x multiply-adds)   share")
for (lv, kind), (cnt, smax, fmax, m) in sorted(by_level.items()):
    print(f" {lv:5d} {kind:5s} {cnt:6d}  {smax:10d}  {fmax:9d}   {2 * m / 1e6:10.2f}   {100 * m / tot:5.1f} %")
banded = 2.0 * 36 * 16 ** 3 * 8 * n * (n / 128) ** 3  # k_press128s: 36 tiles x 8 panels of rank-16 updates per block, n blocks (128-wide formula scaled)
print(f"factorisation: {2 * tot / 1e6:.1f} Mflop per member; block elimination along ix (k_press128s accounting): {banded / 1e6:.1f} Mflop -> {banded / (2 * tot):.1f} x fewer")
print(f"substitution: {2 * solve / 1e6:.2f} Mflop per right-hand side; factor {fac_mem / 1e6:.2f} MB per member (block elimination: {n * 36 * 256 * 8 * (n / 128) ** 2 / 1e6:.2f} MB), "
      f"update matrices if all kept {upd_mem / 1e6:.2f} MB (a stack holds one branch)")
big = sorted(fronts, key=lambda f: -(f[1] + f[2]))[:6]
print("largest fronts (level, pivots, boundary):", [(lv, s, b) for lv, s, b, _ in big])
