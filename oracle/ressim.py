"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the forward simulator.

PARITY UNPINNED.  The arithmetic of this half of the hot path lives in the third-party
package ``TPFA-ResSim`` pinned at ``git@adc89536`` (reference ``requirements.txt:1``),
imported as ``import TPFA_ResSim as simulator`` (``notebooks/HistoryMatch.py:88``).  Its
source is absent from ``/root/reference`` and it is not installable here, and the
reference holds no golden vector, stored notebook output or test for a single simulator
number.  This file therefore restates the *published* algorithm the reference cites
(``HistoryMatch.py:93-95``: Aarnes, Gimse & Lie, "An introduction to the numerics of flow
in porous media using Matlab", listings TPFA / RelPerm / Pres / GenA / Upstream) in the
C-ordered ``(Nx, Ny)`` layout the reference's call sites imply, and
anchors the API on the reference's call sites:

    simulator.ResSim(Nx, Ny, Lx, Ly)            HistoryMatch.py:97, Optimise.py:64
    model.K = stack([p, p])                     HistoryMatch.py:160-164
    model.prd_xy / inj_xy / inj_rates / ...     HistoryMatch.py:186-190
    model.xy2ind(*model.prd_xy.T)               HistoryMatch.py:209
    model.sim(dt, nTime, wsat0, pbar=False)     HistoryMatch.py:224, 362
    model.ind2xy(...)                           HistoryMatch.py:700-701

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  It uses NumPy + ``scipy.sparse`` exactly as the upstream package does
(sparse 5-diagonal matrices, ``spsolve`` direct solve, explicit upwind sub-cycling), so it
is also the CPU baseline timed next to the MI355X numbers.
"""

from __future__ import annotations

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import spsolve


class ResSim:
    """2-D two-phase incompressible immiscible TPFA simulator (CPU restatement)."""

    def __init__(self, Nx, Ny, Lx=1.0, Ly=1.0, name=""):
        self.Nx, self.Ny, self.Lx, self.Ly, self.name = int(Nx), int(Ny), float(Lx), float(Ly), name
        # --- Grid ---
        self.shape = (self.Nx, self.Ny)
        self.Nxy = self.Nx * self.Ny
        self.hx, self.hy = self.Lx / self.Nx, self.Ly / self.Ny
        self.h2 = self.hx * self.hy
        self.domain = ((0.0, 0.0), (self.Lx, self.Ly))  # Optimise.py:465 uses domain[1]
        xc = (np.arange(self.Nx) + 0.5) * self.hx
        yc = (np.arange(self.Ny) + 0.5) * self.hy
        self.mesh = tuple(np.meshgrid(xc, yc, indexing="ij"))
        # --- Gridded + fluid params (upstream defaults; reference never sets them) ---
        self._K = np.ones((2, self.Nx, self.Ny))
        self.por = np.ones(self.shape)
        self.vw, self.vo, self.swc, self.sor = 1.0, 1.0, 0.0, 0.0
        # --- Wells ---
        self._inj_xy = np.zeros((0, 2))
        self._prd_xy = np.zeros((0, 2))
        self.inj_rates = np.zeros((0, 1))
        self.prd_rates = np.zeros((0, 1))
        self.actual_rates = {}

    # ------------------------------------------------------------------ grid helpers
    def sub2ind(self, ix, iy):
        return np.asarray(ix) * self.Ny + np.asarray(iy)

    def ind2sub(self, ind):
        ind = np.asarray(ind)
        return ind // self.Ny, ind % self.Ny

    def xy2sub(self, x, y):
        x, y = np.asarray(x, float), np.asarray(y, float)
        if np.any((x < 0) | (x > self.Lx) | (y < 0) | (y > self.Ly)):
            raise ValueError("Point outside of the domain")  # Optimise.py:549-554
        ix = (x / self.Lx * self.Nx).astype(int).clip(max=self.Nx - 1)
        iy = (y / self.Ly * self.Ny).astype(int).clip(max=self.Ny - 1)
        return ix, iy

    def xy2ind(self, x, y):
        return self.sub2ind(*self.xy2sub(x, y))

    def sub2xy(self, ix, iy):
        x = (np.asarray(ix) + 0.5) * self.hx
        y = (np.asarray(iy) + 0.5) * self.hy
        return np.array([x, y])

    def ind2xy(self, ind):
        return self.sub2xy(*self.ind2sub(ind))

    # ------------------------------------------------------------------ parameters
    @property
    def K(self):
        return self._K

    @K.setter
    def K(self, val):
        """Accepts (2,Nx,Ny) HistoryMatch.py:164, (1,Nxy) Optimise.py:69, (Nxy,) Optimise.py:888."""
        val = np.asarray(val, float)
        if val.shape == (2, self.Nx, self.Ny):
            self._K = val.copy()
        else:
            p = val.reshape(self.shape)
            self._K = np.stack([p, p])

    @property
    def inj_xy(self):
        return self._inj_xy

    @inj_xy.setter
    def inj_xy(self, v):
        self._inj_xy = self._collocate(v)

    @property
    def prd_xy(self):
        return self._prd_xy

    @prd_xy.setter
    def prd_xy(self, v):
        self._prd_xy = self._collocate(v)

    def _collocate(self, xy):
        """Wells are collocated to cell centres (HistoryMatch.py:197)."""
        xy = np.asarray(xy, float).reshape(-1, 2)
        return self.sub2xy(*self.xy2sub(xy[:, 0], xy[:, 1])).T

    @property
    def nInj(self):
        return len(self._inj_xy)

    @property
    def nPrd(self):
        return len(self._prd_xy)

    # ------------------------------------------------------------------ numerics
    def source_field(self, k):
        """q[inj] += rate, q[prd] -= rate; column k, or 0 if a single column."""
        q = np.zeros(self.Nxy)
        inj = np.asarray(self.inj_rates, float).reshape(self.nInj, -1)
        prd = np.asarray(self.prd_rates, float).reshape(self.nPrd, -1)
        ri = inj[:, k if inj.shape[1] > 1 else 0]
        rp = prd[:, k if prd.shape[1] > 1 else 0]
        if not np.isclose(ri.sum(), rp.sum()):
            raise ValueError("Sum of injection rates must equal sum of production rates")  # HM.py:182-184
        np.add.at(q, self.xy2ind(*self._inj_xy.T), ri)
        np.add.at(q, self.xy2ind(*self._prd_xy.T), -rp)
        return q, ri, rp

    def rel_perm(self, s):
        """Listing RelPerm: quadratic mobilities (rel.perm / viscosity)."""
        S = (s - self.swc) / (1 - self.swc - self.sor)
        Mw = S**2 / self.vw
        Mo = (1 - S) ** 2 / self.vo
        return Mw, Mo

    def spdiags(self, data, diags):
        """5-diagonal (Nxy,Nxy) matrix; scipy convention: A[j-k, j] = data_k[j]."""
        return sparse.spdiags(data, diags, self.Nxy, self.Nxy)

    def tpfa(self, KM, q):
        """Listing TPFA. KM: (2,Nx,Ny). Returns P (Nx,Ny), Vx (Nx+1,Ny), Vy (Nx,Ny+1)."""
        Nx, Ny, hx, hy = self.Nx, self.Ny, self.hx, self.hy
        L = KM ** (-1)
        TX = np.zeros((Nx + 1, Ny))
        TY = np.zeros((Nx, Ny + 1))
        TX[1:-1, :] = 2 * hy / hx / (L[0, :-1, :] + L[0, 1:, :])
        TY[:, 1:-1] = 2 * hx / hy / (L[1, :, :-1] + L[1, :, 1:])
        x1 = TX[:-1, :].ravel()
        x2 = TX[1:, :].ravel()
        y1 = TY[:, :-1].ravel()
        y2 = TY[:, 1:].ravel()
        diag = y1 + y2 + x1 + x2
        diag[0] += np.sum(self._K[:, 0, 0])  # coerce SPD (fixes the additive constant)
        A = self.spdiags([-x2, -y2, diag, -y1, -x1], [-Ny, -1, 0, 1, Ny])
        u = spsolve(A.tocsr(), q)
        P = u.reshape(self.shape)
        Vx = np.zeros((Nx + 1, Ny))
        Vy = np.zeros((Nx, Ny + 1))
        Vx[1:-1, :] = (P[:-1, :] - P[1:, :]) * TX[1:-1, :]
        Vy[:, 1:-1] = (P[:, :-1] - P[:, 1:]) * TY[:, 1:-1]
        return P, Vx, Vy

    def pressure_step(self, S, q):
        """Listing Pres: mobility-weighted permeability, then TPFA."""
        Mw, Mo = self.rel_perm(S)
        Mt = (Mw + Mo).reshape(self.shape)
        KM = Mt * self._K
        return self.tpfa(KM, q)

    def upwind_diff(self, Vx, Vy, q):
        """Listing GenA: upwind flux matrix."""
        fp = q.clip(max=0)
        x1 = Vx.clip(max=0)[:-1, :].ravel()
        x2 = Vx.clip(min=0)[1:, :].ravel()
        y1 = Vy.clip(max=0)[:, :-1].ravel()
        y2 = Vy.clip(min=0)[:, 1:].ravel()
        return self.spdiags([x2, y2, fp + x1 - x2 + y1 - y2, -y1, -x1], [-self.Ny, -1, 0, 1, self.Ny])

    def cfl_substeps(self, Vx, Vy, q, T):
        """number of explicit sub-steps and the per-cell local step dtx."""
        pv = self.h2 * self.por.ravel()
        fi = q.clip(min=0)
        XP, XN = Vx.clip(min=0), Vx.clip(max=0)
        YP, YN = Vy.clip(min=0), Vy.clip(max=0)
        Vi = XP[:-1] + YP[:, :-1] - XN[1:] - YN[:, 1:]
        with np.errstate(divide="ignore"):
            pm = min(pv / (Vi.ravel() + fi))
        sat = self.swc + self.sor
        cfl = ((1 - sat) / 3) * pm
        Nts = int(np.ceil(T / cfl))
        dtx = (T / Nts) / pv
        if getattr(self, "_trace", None) is not None:  # test diagnostics: the sub-step count per time step and its argument
            self._trace.append((Nts, T / cfl))
        return Nts, dtx, fi

    def saturation_step_upwind(self, S, q, Vx, Vy, T):
        """Listing Upstream: explicit upwind, sub-cycled under the CFL limit."""
        Nts, dtx, fi = self.cfl_substeps(Vx, Vy, q, T)
        A = self.upwind_diff(Vx, Vy, q)
        A = sparse.spdiags(dtx, 0, self.Nxy, self.Nxy) @ A
        for _ in range(Nts):
            mw, mo = self.rel_perm(S)
            fw = mw / (mw + mo)
            S = S + (A @ fw + fi * dtx)
        return S

    def step(self, S, k, dt):
        q, ri, rp = self.source_field(k)
        _, Vx, Vy = self.pressure_step(S, q)
        return self.saturation_step_upwind(S, q, Vx, Vy, dt), ri, rp

    def sim(self, dt, nTime, wsat0, pbar=False):
        """SURVEY.md A.5. Returns (nTime+1, Nxy); row 0 = wsat0 (HistoryMatch.py:224-225)."""
        wsats = np.zeros((nTime + 1, self.Nxy))
        wsats[0] = wsat0
        inj, prd = [], []
        self._trace = []  # (Nts, dt / cfl) per time step: `nts_trace` after the run
        for k in range(nTime):
            wsats[k + 1], ri, rp = self.step(wsats[k], k, dt)
            inj.append(ri)
            prd.append(rp)
        self.nts_trace = np.array([t[0] for t in self._trace], dtype=np.int64)
        self.cfl_arg_trace = np.array([t[1] for t in self._trace])
        self._trace = None
        self.actual_rates = dict(inj=np.array(inj).T.reshape(self.nInj, -1),
                                 prd=np.array(prd).T.reshape(self.nPrd, -1))
        return wsats

    # ---- stencil (per-cell) form of the saturation sub-step: what the HIP kernel evaluates ----
    def saturation_step_stencil(self, S, q, Vx, Vy, T):
        """Same update as `saturation_step_upwind`, written cell-wise with the summation order of a
        CSR row that `spdiags(dtx) @ A` yields in SciPy 1.15 -- DESCENDING column indices
        (E=j+Ny, N=j+1, C=j, S=j-1, W=j-Ny):

            S_c <- S_c + (((((cE*fE + cN*fN) + cC*fC) + cS*fS) + cW*fW) + fi_c*dtx_c)

        with c* = dtx_c * (unscaled coefficient), each product rounded once (no FMA).
        tests/ check this against the sparse-matrix form bit-for-bit."""
        Nx, Ny = self.shape
        Nts, dtx, fi = self.cfl_substeps(Vx, Vy, q, T)
        fp = q.clip(max=0).reshape(Nx, Ny)
        d = dtx.reshape(Nx, Ny)
        XN, XP = Vx.clip(max=0), Vx.clip(min=0)
        YN, YP = Vy.clip(max=0), Vy.clip(min=0)
        x1, x2, y1, y2 = XN[:-1, :], XP[1:, :], YN[:, :-1], YP[:, 1:]
        cC = d * (fp + x1 - x2 + y1 - y2)
        cW = d * XP[:-1, :]   # row j, col j-Ny : x2[j-Ny] = XP[ix, iy]
        cE = d * (-XN[1:, :])  # row j, col j+Ny : -x1[j+Ny] = -XN[ix+1, iy]
        cS = d * YP[:, :-1]   # row j, col j-1  : y2[j-1] = YP[ix, iy]
        cN = d * (-YN[:, 1:])  # row j, col j+1  : -y1[j+1] = -YN[ix, iy+1]
        fid = (fi * dtx).reshape(Nx, Ny)
        S = S.reshape(Nx, Ny).copy()
        for _ in range(Nts):
            mw, mo = self.rel_perm(S)
            f = mw / (mw + mo)
            fpad = np.zeros((Nx + 2, Ny + 2))
            fpad[1:-1, 1:-1] = f
            acc = cE * fpad[2:, 1:-1]
            acc = acc + cN * fpad[1:-1, 2:]
            acc = acc + cC * f
            acc = acc + cS * fpad[1:-1, :-2]
            acc = acc + cW * fpad[:-2, 1:-1]
            S = S + (acc + fid)
        return S.ravel()

    # ---- the build's fp32 forward mode (NOT in the reference, which is fp64 end to end: HistoryMatch.py:362) ----
    F32_FOLD = 64
    F32_HOT = 0.9997  # sat32.h: cells entering a time step at S >= this get the saturated-neighbourhood rule (diag32)

    def saturation_step_stencil_f32c(self, S32, q, Vx, Vy, T, compensated=True):
        """Specification of the saturation step of ``dtype=32`` plans (historymatching_amd/csrc/sat32.h), operation for
        operation in NumPy float32 -- the HIP kernels are compared with it bit for bit.  The reference has no such mode; its
        accuracy against `saturation_step_upwind` is what the parity tests bound (<= 1e-3 on S over a whole run).

        Pressure, fluxes, the CFL bound, `Nts` and `d = dtx` are fp64 (as in the fp64 mode: the sub-step count is a
        discontinuity).  The five coefficients and `fi*d` are formed in fp64 and rounded to float32 ONCE; the fractional flow,
        its five products and their sum are float32.  The saturation is carried as a float32 pair (base, dS), `S = base + dS`:
        a sub-step adds its increment to dS only, whose ulp is that of the change since the last fold, not that of S -- a plain
        float32 accumulator loses every increment below half an ulp of S (2.6e-4 / 2.3e-3 / > 1e-2 of drift at 128^2 / 256^2 /
        512^2 over 40 steps, profiles/r05/fp32_drift_*_before.txt).  Every `F32_FOLD` sub-steps dS is folded into base by an
        exact two-sum (the rounding error stays in dS); the stored state of a time step is `base + dS` rounded once.
        `compensated=False`: the plain float32 accumulator of rounds 1-4 (kept to measure what the pair buys)."""
        f32 = np.float32
        Nx, Ny = self.shape
        Nts, dtx, fi = self.cfl_substeps(Vx, Vy, q, T)
        fp = q.clip(max=0).reshape(Nx, Ny)
        d = dtx.reshape(Nx, Ny)
        XN, XP = Vx.clip(max=0), Vx.clip(min=0)
        YN, YP = Vy.clip(max=0), Vy.clip(min=0)
        x1, x2, y1, y2 = XN[:-1, :], XP[1:, :], YN[:, :-1], YP[:, 1:]
        cC64 = d * (fp + x1 - x2 + y1 - y2)
        cC = cC64.astype(f32)
        cW = (d * XP[:-1, :]).astype(f32)
        cE = (d * (-XN[1:, :])).astype(f32)
        cS = (d * YP[:, :-1]).astype(f32)
        cN = (d * (-YN[:, 1:])).astype(f32)
        # the source term of an injector's cell is rounded JOINTLY with its diagonal coefficient (sat32.h: source32): fid = fl32((c_C + fi d)
        # - cC32), so that cC32 + fid is the fp64 sum to one rounding -- zero for a pure source cell, which then fills up to S = 1 and stays
        # (0 <= S <= 1); rounded independently the two differ by an ulp that the cell gains every sub-step at fw = 1
        fid64 = (fi * dtx).reshape(Nx, Ny)
        fid = np.where(fi.reshape(Nx, Ny) > 0, (cC64 + fid64) - cC.astype(np.float64), fid64).astype(f32)
        # ... and for the cells that enter the step at S >= F32_HOT c_C is lowered by whole ulps until the increment of a SATURATED
        # neighbourhood (every fw = 1), evaluated in float32 in the sweeps' order of additions, is not positive (sat32.h: diag32): in exact
        # arithmetic that sum is d (q - div V) = 0, rounded one by one it is an ulp-sized residue, the same in all Nts sub-steps of a
        # step, on which cells around the injector creep above 1
        hot = np.asarray(S32, dtype=f32).reshape(Nx, Ny) >= f32(self.F32_HOT)
        for _ in range(4):
            r = ((((cE + cN) + cC) + cS) + cW) + fid
            pos = hot & (r > 0)
            if not pos.any():
                break
            cC = np.where(pos, np.nextafter(cC, f32(-np.inf)), cC)
        assert cC.dtype == f32
        base = np.asarray(S32, dtype=f32).reshape(Nx, Ny).copy()
        dS = np.zeros((Nx, Ny), dtype=f32)
        one = f32(1)
        default = self.vw == 1 and self.vo == 1 and self.swc == 0 and self.sor == 0
        fpad = np.zeros((Nx + 2, Ny + 2), dtype=f32)
        for it in range(Nts):
            s = base + dS
            if default:
                mw = s * s
                o = one - s
                mo = o * o
            else:  # rel_perm<float> of fwd_dev.h
                den = f32((1.0 - self.swc) - self.sor)
                Sn = (s - f32(self.swc)) / den
                mw = (Sn * Sn) / f32(self.vw)
                o = one - Sn
                mo = (o * o) / f32(self.vo)
            f = mw / (mw + mo)
            fpad[1:-1, 1:-1] = f
            acc = cE * fpad[2:, 1:-1]
            acc = acc + cN * fpad[1:-1, 2:]
            acc = acc + cC * f
            acc = acc + cS * fpad[1:-1, :-2]
            acc = acc + cW * fpad[:-2, 1:-1]
            if not compensated:
                base = base + (acc + fid)
                continue
            dS = dS + (acc + fid)
            if (it & (self.F32_FOLD - 1)) == self.F32_FOLD - 1:  # exact two-sum: base + dS == t + e
                t = base + dS
                bb = t - base
                dS = (base - (t - bb)) + (dS - bb)
                base = t
            assert dS.dtype == f32 and base.dtype == f32
        return (base + dS).ravel()

    def sim_f32c(self, dt, nTime, wsat0, compensated=True):
        """`sim` in the build's fp32 mode: fp64 pressure step on the float32 state, `saturation_step_stencil_f32c`."""
        wsats = np.zeros((nTime + 1, self.Nxy), dtype=np.float32)
        wsats[0] = wsat0
        self._trace = []
        for k in range(nTime):
            q, _, _ = self.source_field(k)
            _, Vx, Vy = self.pressure_step(wsats[k].astype(np.float64), q)
            wsats[k + 1] = self.saturation_step_stencil_f32c(wsats[k], q, Vx, Vy, dt, compensated)
        self.nts_trace = np.array([t[0] for t in self._trace], dtype=np.int64)
        self._trace = None
        return wsats


def perm_transf(x):
    """HistoryMatch.py:137-138."""
    return 0.1 + np.exp(5 * x)


def set_perm(model, log_perm_array):
    """HistoryMatch.py:160-164."""
    p = perm_transf(log_perm_array).reshape(model.shape)
    model.K = np.stack([p, p])


def default_wells(model):
    """Well layout of HistoryMatch.py:177-190: 4 producers near the corners, 1 central injector."""
    near01 = np.array([0.12, 0.87])
    xy_4corners = [[x, y] for y in model.Ly * near01 for x in model.Lx * near01]
    model.prd_xy = xy_4corners
    model.inj_xy = [[model.Lx / 2, model.Ly / 2]]
    model.inj_rates = [[1]]
    model.prd_rates = np.ones((4, 1)) / 4
    return model


def comp1(model, perm, wsat0, dt, nTime):
    """HistoryMatch.py:358-364 (the reference deep-copies a module-global model)."""
    import copy

    new_model = copy.deepcopy(model)
    set_perm(new_model, perm)
    wsats = new_model.sim(dt, nTime, wsat0, pbar=False)
    prod_inds = new_model.xy2ind(*new_model.prd_xy.T)
    prods = np.array([x[prod_inds] for x in wsats[1:]])
    return wsats, prods


def _comp1_star(args):
    return comp1(*args)


def make_pool(nproc):
    """Process pool as the reference uses it (utils.py:201-224): one worker per core, BLAS pinned to one thread."""
    import multiprocessing as mp

    import threadpoolctl

    threadpoolctl.threadpool_limits(1)
    return mp.get_context("fork").Pool(nproc)


def forward_model(model, perms, wsat0s=None, dt=0.025, nTime=40, nproc=1, pool=None):
    """HistoryMatch.py:383-387 + utils.apply (utils.py:155-242): ordered map over members,
    one process per core with BLAS pinned to one thread (utils.py:201-224); stdlib
    multiprocessing stands in for pathos.  `pool` reuses an existing `make_pool` (benchmarks)."""
    perms = np.asarray(perms)
    if wsat0s is None:
        wsat0s = np.zeros((len(perms), model.Nxy))
    if len(wsat0s) != len(perms):
        raise ValueError("zip() argument 2 is shorter/longer than argument 1")  # utils.py:175 strict zip
    tasks = [(model, p, w, dt, nTime) for p, w in zip(perms, wsat0s)]
    if pool is not None:
        output = pool.map(_comp1_star, tasks, chunksize=1)
    elif nproc > 1:
        with make_pool(nproc) as own:
            output = own.map(_comp1_star, tasks, chunksize=1)
    else:
        output = [_comp1_star(t) for t in tasks]
    return [np.asarray(y) for y in zip(*output)]
