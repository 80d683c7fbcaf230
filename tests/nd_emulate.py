"""NumPy emulation of the nested-dissection pressure solve exactly as press_nd.hip's kernels read the symbolic tables
(hm_debug_nd_tables): fronts in tiles of 16 with identity-padded pivots, the right-hand side as an extra boundary row,
children's update matrices packed lower-triangular and gathered through `cpos`, the factor as W^T tiles, back substitution
x1 = -W^T [x2; -1].  Test infrastructure (CPU): it checks the TABLES and the algebra the kernels implement, not the kernels."""
import ctypes as C

import numpy as np

F_LEVEL, F_S, F_B, F_ST, F_BT, F_C0, F_C1, F_CELLS, F_FACT, F_UPD, F_KREG, F_BC0, F_BC1, F_PBOX, F_RBOX, F_REC = range(16)


def tables(lib, Nx, Ny):
    info = (C.c_longlong * 64)()
    rc = lib.hm_debug_nd_tables(Nx, Ny, info, None, None, None, None)
    if rc:
        raise RuntimeError(lib.hm_last_error().decode())
    nF, nC = int(info[0]), int(info[1])
    fronts = np.zeros((nF, int(info[19])), dtype=np.int32)
    cells = np.zeros(nC, dtype=np.int32)
    cpos = np.zeros(2 * nC, dtype=np.int16)
    rec = np.zeros(256 * int(info[7]), dtype=np.int16)
    rc = lib.hm_debug_nd_tables(Nx, Ny, info, fronts.ctypes.data_as(C.POINTER(C.c_int)), cells.ctypes.data_as(C.POINTER(C.c_int)),
                                cpos.ctypes.data_as(C.POINTER(C.c_short)), rec.ctypes.data_as(C.POINTER(C.c_short)))
    assert rc == 0
    return dict(info=[int(v) for v in info], fronts=fronts, cells=cells, cpos=cpos, rec=rec.reshape(-1, 64, 4), Nx=Nx, Ny=Ny)


def coefficient(tab, dg, TX, TY, cm, ck):
    """A[cm, ck] of the five-point system the way the kernels form it from the face arrays (flat TX (Nx+1)*Ny, TY Nx*(Ny+1))."""
    Ny = tab["Ny"]
    d = cm - ck
    ty = ck + ck // Ny
    if d == 0:
        return dg[ck]
    if d == Ny:
        return -TX[ck + Ny]
    if d == -Ny:
        return -TX[ck]
    if d == 1:
        return -TY[ty + 1]
    if d == -1:
        return -TY[ty]
    return 0.0


def matrix(tab, dg, TX, TY):
    """The five-point system as a CSR matrix, entry by entry through `coefficient`'s rules (vectorised)."""
    import scipy.sparse as sp

    Nx, Ny = tab["Nx"], tab["Ny"]
    c = np.arange(Nx * Ny)
    ty = c + c // Ny
    rows = [c, c[Ny:], c[:-Ny], c[c % Ny != Ny - 1] + 1, c[c % Ny != 0] - 1]
    cols = [c, c[Ny:] - Ny, c[:-Ny] + Ny, c[c % Ny != Ny - 1], c[c % Ny != 0]]
    # A[cm, ck]: d = cm - ck = Ny -> -TX[ck + Ny]; -Ny -> -TX[ck]; 1 -> -TY[ty + 1]; -1 -> -TY[ty]
    vals = [dg[c], -TX[cols[1] + Ny], -TX[cols[2]], -TY[ty[cols[3]] + 1], -TY[ty[cols[4]]]]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(Nx * Ny, Nx * Ny))


def tri(a, b):
    hi, lo = (a, b) if a >= b else (b, a)
    return hi * (hi + 1) // 2 + lo


def solve(tab, dg, TX, TY, q):
    """Returns (x, factor dict, flops) -- x solves the five-point system with diagonal dg, faces TX/TY, right-hand side q."""
    fronts, cells, cpos = tab["fronts"], tab["cells"], tab["cpos"]
    nF = len(fronts)
    A = matrix(tab, dg, TX, TY)
    upd = {}
    fact = {}
    order = sorted(range(nF), key=lambda f: -fronts[f][F_LEVEL])  # deepest level first: children before parents
    for f in order:
        F = fronts[f]
        s, b, st, bt = int(F[F_S]), int(F[F_B]), int(F[F_ST]), int(F[F_BT])
        T = st + bt
        cl = cells[F[F_CELLS]:F[F_CELLS] + 16 * T]
        kids = [int(F[F_C0]), int(F[F_C1])]
        cp = [cpos[2 * F[F_CELLS] + c * 16 * T: 2 * F[F_CELLS] + (c + 1) * 16 * T] for c in range(2)]
        n = 16 * T
        # the whole (padded) front as a dense symmetric matrix; only pivot columns get A entries
        M = np.zeros((n, n))
        piv = cl[:16 * st]
        pk = np.nonzero(piv >= 0)[0]
        rows = np.nonzero(cl >= 0)[0]
        if len(pk):
            blk = A[cl[rows]][:, piv[pk]].toarray()
            M[np.ix_(rows, pk)] = blk
            M[np.ix_(pk, rows)] = blk.T
            rhs = np.nonzero(cl == -2)[0]
            M[np.ix_(rhs, pk)] = q[piv[pk]][None, :]
            M[np.ix_(pk, rhs)] = q[piv[pk]][:, None]
        pad = np.nonzero(piv < 0)[0]
        M[pad, pad] = 1.0
        if kids[0] >= 0:
            for c in range(2):
                U = upd.pop(kids[c])
                p = cp[c]
                idx = np.nonzero(p >= 0)[0]
                pi = p[idx].astype(np.int64)
                hi, lo = np.maximum(pi[:, None], pi[None, :]), np.minimum(pi[:, None], pi[None, :])
                add = U[hi * (hi + 1) // 2 + lo]
                rhs = cl[idx] == -2
                add[np.ix_(rhs, rhs)] = 0.0  # the (rhs, rhs) entry is never used
                M[np.ix_(idx, idx)] += add
        # panel elimination, tile by tile
        WT = {}
        for p_ in range(st):
            sl = slice(16 * p_, 16 * p_ + 16)
            P = np.linalg.inv(M[sl, sl])
            rest = slice(16 * p_ + 16, n)
            W = M[rest, sl] @ P            # rows below x 16 pivots
            WT[p_] = W.T.copy()            # 16 x rows: what the kernels store
            M[rest, rest] -= W @ M[rest, sl].T
        fact[f] = WT
        if b > 0 or True:
            bb = b + 1
            Ub = M[16 * st:16 * st + bb, 16 * st:16 * st + bb]
            packed = np.zeros(bb * (bb + 1) // 2)
            for i in range(bb):
                packed[i * (i + 1) // 2:i * (i + 1) // 2 + i + 1] = Ub[i, :i + 1]
            upd[f] = packed
    # back substitution, root first
    x = np.zeros(tab["Nx"] * tab["Ny"])
    for f in sorted(range(nF), key=lambda f: fronts[f][F_LEVEL]):
        F = fronts[f]
        st, bt = int(F[F_ST]), int(F[F_BT])
        T = st + bt
        cl = cells[F[F_CELLS]:F[F_CELLS] + 16 * T]
        xe = np.where(cl >= 0, x[np.maximum(cl, 0)], np.where(cl == -2, -1.0, 0.0))
        for p_ in reversed(range(st)):
            x1 = -fact[f][p_] @ xe[16 * p_ + 16:]
            xe[16 * p_:16 * p_ + 16] = x1
            for k in range(16):
                if cl[16 * p_ + k] >= 0:
                    x[cl[16 * p_ + k]] = x1[k]
    return x
