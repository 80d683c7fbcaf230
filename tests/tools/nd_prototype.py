"""Prototype of the nested-dissection (multifrontal) pressure solve that DESIGN.md section 8 names as the next kernel -- NumPy, one
member, to fix the ordering, the front index sets and the extend-add maps before any of it is written for the GPU, and to check
the whole against a sparse direct solve of the same TPFA system.

    python tests/tools/nd_prototype.py [n=128] [leaf=4] [seed=1]

What it establishes (printed at the end):
  * the fronts of the geometric dissection (one-cell separators across the longer side of a region, down to leaf x leaf boxes) are
    exactly `pivots + the region's sides that are ancestor separators` -- no corner cells -- and every child's boundary lies inside
    its parent's front, so the extend-add is a gather through a per-front index list that depends on the grid only (built once on
    the host, shared by all members);
  * the solution agrees with scipy.sparse.linalg.spsolve to the solve's own rounding;
  * flops, front sizes and the length of the pivot chain as profiles/tools/nd_flops.py counts them.

The TPFA system is the one of SURVEY.md Appendix A.3: harmonic-mean face transmissibilities of K * lambda(S), zero on the boundary,
`A[0,0] += Kx[0,0] + Ky[0,0]`; here with S = 0 (lambda = 1/vo = 1) and a log-normal K, wells in opposite corners."""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.linalg import cholesky, solve_triangular

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
Nx = Ny = n


# ---------------------------------------------------------------- the system
def tpfa(K):
    """Face transmissibilities TX (Nx+1, Ny), TY (Nx, Ny+1) and the five-point matrix in CSR form (cell index ix * Ny + iy)."""
    L = 1.0 / K
    TX = np.zeros((Nx + 1, Ny))
    TY = np.zeros((Nx, Ny + 1))
    TX[1:-1] = 2.0 / (L[:-1] + L[1:])
    TY[:, 1:-1] = 2.0 / (L[:, :-1] + L[:, 1:])
    dg = TX[:-1] + TX[1:] + TY[:, :-1] + TY[:, 1:]
    dg[0, 0] += 2.0 * K[0, 0]
    idx = np.arange(Nx * Ny).reshape(Nx, Ny)
    rows = [idx.ravel(), idx[1:].ravel(), idx[:-1].ravel(), idx[:, 1:].ravel(), idx[:, :-1].ravel()]
    cols = [idx.ravel(), idx[:-1].ravel(), idx[1:].ravel(), idx[:, :-1].ravel(), idx[:, 1:].ravel()]
    vals = [dg.ravel(), -TX[1:-1].ravel(), -TX[1:-1].ravel(), -TY[:, 1:-1].ravel(), -TY[:, 1:-1].ravel()]
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(Nx * Ny, Nx * Ny))
    return TX, TY, dg, A


# ---------------------------------------------------------------- symbolic phase (grid only: shared by every member)
class Front:
    __slots__ = ("piv", "bnd", "children", "level", "kind", "child_maps", "L11", "L21", "y1")


def cells(x0, x1, y0, y1):
    return (np.arange(x0, x1)[:, None] * Ny + np.arange(y0, y1)[None, :]).ravel()


def dissect(x0, x1, y0, y1, sides, level, order):
    """Region [x0, x1) x [y0, y1); sides = (W, E, S, N): that side borders an ancestor's separator.  Appends the region's fronts
    to `order` in post-order (children before parents) and returns the region's top front."""
    f = Front()
    f.level, f.children = level, []
    bnd = []
    if sides[0]:
        bnd.append(cells(x0 - 1, x0, y0, y1))
    if sides[1]:
        bnd.append(cells(x1, x1 + 1, y0, y1))
    if sides[2]:
        bnd.append(cells(x0, x1, y0 - 1, y0))
    if sides[3]:
        bnd.append(cells(x0, x1, y1, y1 + 1))
    f.bnd = np.concatenate(bnd) if bnd else np.zeros(0, dtype=np.int64)
    w, h = x1 - x0, y1 - y0
    if w <= leaf and h <= leaf:
        f.kind, f.piv = "leaf", cells(x0, x1, y0, y1)
    elif w >= h:
        xs = x0 + (w - 1) // 2
        f.kind, f.piv = "sep", cells(xs, xs + 1, y0, y1)
        if xs > x0:
            f.children.append(dissect(x0, xs, y0, y1, (sides[0], True, sides[2], sides[3]), level + 1, order))
        if x1 > xs + 1:
            f.children.append(dissect(xs + 1, x1, y0, y1, (True, sides[1], sides[2], sides[3]), level + 1, order))
    else:
        ys = y0 + (h - 1) // 2
        f.kind, f.piv = "sep", cells(x0, x1, ys, ys + 1)
        if ys > y0:
            f.children.append(dissect(x0, x1, y0, ys, (sides[0], sides[1], sides[2], True), level + 1, order))
        if y1 > ys + 1:
            f.children.append(dissect(x0, x1, ys + 1, y1, (sides[0], sides[1], True, sides[3]), level + 1, order))
    order.append(f)
    return f


order = []
t0 = time.perf_counter()
root = dissect(0, Nx, 0, Ny, (False, False, False, False), 0, order)
local = np.full(Nx * Ny, -1, dtype=np.int64)
for f in order:  # extend-add maps: position of each child boundary cell in the parent's front (pivots first, then boundary)
    front = np.concatenate([f.piv, f.bnd])
    local[front] = np.arange(len(front))
    f.child_maps = []
    for c in f.children:
        m = local[c.bnd]
        assert (m >= 0).all(), "a child's boundary must lie inside its parent's front"
        f.child_maps.append(m)
    local[front] = -1
t_symbolic = time.perf_counter() - t0
assert sorted(np.concatenate([f.piv for f in order]).tolist()) == list(range(Nx * Ny)), "every cell is a pivot exactly once"

# ---------------------------------------------------------------- numeric phase (per member)
rng = np.random.RandomState(seed)
z = rng.randn(Nx, Ny)
for _ in range(6):  # a smooth-ish log-normal field (the prototype needs no particular covariance)
    z = (z + np.roll(z, 1, 0) + np.roll(z, -1, 0) + np.roll(z, 1, 1) + np.roll(z, -1, 1)) / 5
K = 0.1 + np.exp(5 * z / z.std())
TX, TY, dg, A = tpfa(K)
q = np.zeros(Nx * Ny)
q[(Nx // 2) * Ny + Ny // 2] = 1.0
q[0] = q[Ny - 1] = q[(Nx - 1) * Ny] = q[Nx * Ny - 1] = -0.25


def entry(i, j):
    """A[i, j] for cell arrays i (column vector) and j (row vector), from the face arrays -- what a kernel would assemble."""
    ix, iy, jx, jy = i // Ny, i % Ny, j // Ny, j % Ny
    out = np.where((ix == jx) & (iy == jy), dg[ix, iy], 0.0)
    out = out - np.where((jx == ix + 1) & (iy == jy), TX[np.minimum(ix + 1, Nx), iy], 0.0) - np.where((jx == ix - 1) & (iy == jy), TX[ix, iy], 0.0)
    out = out - np.where((jy == iy + 1) & (ix == jx), TY[ix, np.minimum(iy + 1, Ny)], 0.0) - np.where((jy == iy - 1) & (ix == jx), TY[ix, iy], 0.0)
    return out


t0 = time.perf_counter()
flops = 0.0
updates = {}
for f in order:
    s, b = len(f.piv), len(f.bnd)
    front = np.concatenate([f.piv, f.bnd])
    F = np.zeros((s + b, s + b))
    F[:s, :] = entry(f.piv[:, None], front[None, :])  # fully summed rows: the pivots' rows of A, over the whole front
    F[s:, :s] = F[:s, s:].T
    for c, m in zip(f.children, f.child_maps):
        F[np.ix_(m, m)] += updates.pop(id(c))          # extend-add
    f.L11 = cholesky(F[:s, :s], lower=True)
    f.L21 = solve_triangular(f.L11, F[:s, s:], lower=True).T if b else np.zeros((0, s))
    if b:
        updates[id(f)] = F[s:, s:] - f.L21 @ f.L21.T
    flops += 2 * (s ** 3 / 3 + s * s * b + s * b * b)
assert not updates
t_factor = time.perf_counter() - t0

t0 = time.perf_counter()
rhs = q.copy()
for f in order:                                          # forward substitution, leaves to root
    f.y1 = solve_triangular(f.L11, rhs[f.piv], lower=True)
    if len(f.bnd):
        rhs[f.bnd] -= f.L21 @ f.y1
x = np.zeros(Nx * Ny)
for f in reversed(order):                                # back substitution, root to leaves
    t = f.y1 - (f.L21.T @ x[f.bnd] if len(f.bnd) else 0.0)
    x[f.piv] = solve_triangular(f.L11, t, lower=True, trans="T")
t_solve = time.perf_counter() - t0

t0 = time.perf_counter()
x_ref = spla.spsolve(A.tocsc(), q)
t_ref = time.perf_counter() - t0
err = np.abs(x - x_ref).max() / np.abs(x_ref).max()
res = np.abs(A @ x - q).max()
res_ref = np.abs(A @ x_ref - q).max()


def chain(f):
    return len(f.piv) + max((chain(c) for c in f.children), default=0)


print(f"{Nx} x {Ny}, leaves <= {leaf} x {leaf}: {len(order)} fronts ({sum(f.kind == 'leaf' for f in order)} leaves), "
      f"largest front {max(len(f.piv) + len(f.bnd) for f in order)} (pivots {max(len(f.piv) for f in order)}), "
      f"extend-add index lists {sum(len(m) for f in order for m in f.child_maps)} entries in all")
print(f"pivots on the longest leaf-to-root path: {chain(root)} (= {chain(root) / 16:.0f} rank-16 steps; block elimination along ix: {Nx * Ny // 16})")
print(f"factorisation {flops / 1e6:.1f} Mflop, factor {sum(f.L11.size + f.L21.size for f in order) * 8 / 1e6:.2f} MB")
print(f"max |x - x_spsolve| / max |x| = {err:.2e}; residual max |A x - q| = {res:.2e} (spsolve: {res_ref:.2e})")
print(f"NumPy wall: symbolic {t_symbolic:.2f} s, factor {t_factor:.2f} s, solve {t_solve:.2f} s; spsolve {t_ref:.2f} s")
assert res <= 4 * res_ref + 1e-12, "the multifrontal solve must be as good a solution of A x = q as the sparse direct one"
