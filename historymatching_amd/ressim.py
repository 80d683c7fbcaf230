"""Host-side mirror of the reference's simulator object, backed by the batched MI355X kernels.

Mirrors the call surface the reference uses from ``TPFA_ResSim.ResSim`` (external package, pinned in
the reference's ``requirements.txt:1``; call sites cited per member below), so that
``notebooks/HistoryMatch.py`` can do ``from historymatching_amd import ressim as simulator`` and keep
every line that touches ``model``.  The arithmetic of ``sim`` runs on the GPU through the C ABI
(``include/hm_abi.h``); this class only holds the grid, the wells and NumPy index helpers.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class ResSim:
    """``simulator.ResSim(Nx, Ny, Lx, Ly[, name])``  (HistoryMatch.py:97, Optimise.py:64)."""

    def __init__(self, Nx, Ny, Lx=1.0, Ly=1.0, name="", dtype=64, device=None):
        self.Nx, self.Ny = int(Nx), int(Ny)
        self.Lx, self.Ly = float(Lx), float(Ly)
        self.name = name
        self.dtype = int(dtype)  # 64 | 32: arithmetic of the saturation sweep (pressure is always fp64)
        self.device = device
        self.hx, self.hy = self.Lx / self.Nx, self.Ly / self.Ny
        self.h2 = self.hx * self.hy
        self.K = np.ones((2, self.Nx, self.Ny))
        self.por = None  # None == porosity 1 everywhere
        self.vw = self.vo = 1.0
        self.swc = self.sor = 0.0
        self._inj_xy = np.zeros((0, 2))
        self._prd_xy = np.zeros((0, 2))
        self.inj_rates = np.zeros((0, 1))
        self.prd_rates = np.zeros((0, 1))
        self.actual_rates = {}
        self.last_stats = None

    # ---------------------------------------------------------------- grid
    @property
    def shape(self):
        return (self.Nx, self.Ny)  # HistoryMatch.py:163

    @property
    def Nxy(self):
        return self.Nx * self.Ny  # HistoryMatch.py:223

    @property
    def domain(self):
        return ((0.0, 0.0), (self.Lx, self.Ly))  # Optimise.py:465

    @property
    def mesh(self):
        """Cell-centre coordinates, ``ij`` indexing (HistoryMatch.py:152, Optimise.py:441)."""
        xs = self.hx * (0.5 + np.arange(self.Nx))
        ys = self.hy * (0.5 + np.arange(self.Ny))
        return tuple(np.meshgrid(xs, ys, indexing="ij"))

    def sub2ind(self, ix, iy):
        return np.ravel_multi_index((np.asarray(ix), np.asarray(iy)), self.shape)

    def ind2sub(self, ind):
        return np.unravel_index(np.asarray(ind), self.shape)

    def xy2sub(self, x, y):
        x = np.asarray(x, dtype=float)
        y = np.asarray(y, dtype=float)
        inside = (0 <= x) & (x <= self.Lx) & (0 <= y) & (y <= self.Ly)
        if not np.all(inside):
            raise ValueError(f"well/point outside the domain [0,{self.Lx}]x[0,{self.Ly}]")  # Optimise.py:549-554
        ix = np.minimum((x / self.Lx * self.Nx).astype(int), self.Nx - 1)
        iy = np.minimum((y / self.Ly * self.Ny).astype(int), self.Ny - 1)
        return ix, iy

    def xy2ind(self, x, y):
        return self.sub2ind(*self.xy2sub(x, y))  # HistoryMatch.py:209

    def sub2xy(self, ix, iy):
        return np.array([self.hx * (np.asarray(ix) + 0.5), self.hy * (np.asarray(iy) + 0.5)])  # plotting.py:326

    def ind2xy(self, ind):
        return self.sub2xy(*self.ind2sub(ind))  # HistoryMatch.py:700-701, 833

    # ---------------------------------------------------------------- parameters
    def __setattr__(self, key, val):
        if key == "K":
            # HistoryMatch.py:164 (2,Nx,Ny); Optimise.py:69 (1,Nxy); Optimise.py:888 (Nxy,)
            val = np.asarray(val, dtype=float)
            if val.shape != (2, self.Nx, self.Ny):
                val = np.broadcast_to(val.reshape(self.shape), (2, self.Nx, self.Ny)).copy()
        object.__setattr__(self, key, val)

    def _snap(self, xy):
        """Wells are collocated with cell centres (HistoryMatch.py:197, Optimise.py:529-531)."""
        xy = np.asarray(xy, dtype=float).reshape(-1, 2)
        return self.sub2xy(*self.xy2sub(xy[:, 0], xy[:, 1])).T

    inj_xy = property(lambda self: self._inj_xy, lambda self, v: setattr(self, "_inj_xy", self._snap(v)))
    prd_xy = property(lambda self: self._prd_xy, lambda self, v: setattr(self, "_prd_xy", self._snap(v)))
    nInj = property(lambda self: len(self._inj_xy))  # Optimise.py:726
    nPrd = property(lambda self: len(self._prd_xy))  # Optimise.py:644

    def _wells(self, nTime):
        inj = np.ascontiguousarray(np.asarray(self.inj_rates, dtype=float).reshape(self.nInj, -1))
        prd = np.ascontiguousarray(np.asarray(self.prd_rates, dtype=float).reshape(self.nPrd, -1))
        for r, nm in ((inj, "inj_rates"), (prd, "prd_rates")):
            if r.shape[1] not in (1, nTime):
                raise ValueError(f"{nm} must have 1 or nTime={nTime} columns, got {r.shape[1]}")
        cols = nTime if max(inj.shape[1], prd.shape[1]) > 1 else 1
        si = np.broadcast_to(inj, (self.nInj, cols)).sum(0)
        sp = np.broadcast_to(prd, (self.nPrd, cols)).sum(0)
        if not np.allclose(si, sp):
            # "If this is not the case, the model will raise an error when run." HistoryMatch.py:182-184
            raise ValueError("total injection rate must equal total production rate")
        inj_ind = np.ascontiguousarray(self.xy2ind(*self._inj_xy.T), dtype=np.int32)
        prd_ind = np.ascontiguousarray(self.xy2ind(*self._prd_xy.T), dtype=np.int32)
        return inj_ind, inj, prd_ind, prd

    # ---------------------------------------------------------------- the batched device call
    def sim_ensemble(self, perms, wsat0s=None, *, dt, nTime, transformed=False, return_history=True):
        """Run ``N`` members at once: the GPU replacement of ``apply(comp1, perms[, wsat0s])``
        (HistoryMatch.py:358-364, 383-387).  ``perms`` is ``(N, Nxy)`` pre-permeability (``transformed=False``:
        ``K = 0.1+exp(5x)`` on device, HistoryMatch.py:137) or permeability itself.
        Returns ``wsats`` ``(N, nTime+1, Nxy)`` (or ``(N, Nxy)`` final state) and ``prods`` ``(N, nTime, nPrd)``."""
        perms = _lib.as_c(perms, np.float64)
        if perms.ndim != 2 or perms.shape[1] != self.Nxy:
            raise ValueError(f"perms must have shape (N, {self.Nxy}), got {perms.shape}")
        N = perms.shape[0]
        ft = np.float64 if self.dtype == 64 else np.float32
        if wsat0s is not None:
            wsat0s = _lib.as_c(wsat0s, ft)
            if wsat0s.shape != (N, self.Nxy):
                # the reference zips ensembles member-wise with strict=True (utils.py:175)
                raise ValueError(f"wsat0s must have shape {(N, self.Nxy)}, got {wsat0s.shape}")
        inj_ind, inj, prd_ind, prd = self._wells(nTime)
        por = None if self.por is None else _lib.as_c(np.asarray(self.por).reshape(-1), np.float64)
        wsats = np.empty((N, nTime + 1, self.Nxy) if return_history else (N, self.Nxy), dtype=ft)
        prods = np.empty((N, nTime, self.nPrd), dtype=ft)
        status = np.zeros(N, dtype=np.int32)
        stats = _lib.hm_stats()
        ctx = _lib.Context.get(self.device)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
        rc = ctx.lib.hm_forward_batched(
            ctx.handle, N, self.Nx, self.Ny, self.Lx, self.Ly, _lib.ptr(perms), int(bool(transformed)),
            _lib.ptr(wsat0s), self.nInj, ip(inj_ind), dp(inj), inj.shape[1], self.nPrd, ip(prd_ind), dp(prd),
            prd.shape[1], float(dt), int(nTime), self.vw, self.vo, self.swc, self.sor,
            None if por is None else dp(por), self.dtype, int(bool(return_history)), _lib.ptr(wsats),
            _lib.ptr(prods), ip(status), C.byref(stats))
        _lib.check(rc, "hm_forward_batched")
        if status.any():
            bad = np.flatnonzero(status)
            raise _lib.HmError(f"forward model failed for members {bad[:8].tolist()} (status {status[bad[:8]].tolist()}): "
                               "1=non-positive pivot in pressure solve, 2=bad CFL, 4=non-finite saturation, "
                               "8=CG pressure solver did not converge")
        self.last_stats = stats.asdict()
        cols = lambda r: np.broadcast_to(r, (r.shape[0], nTime)).copy()  # noqa: E731
        self.actual_rates = dict(inj=cols(inj), prd=cols(prd))  # Optimise.py:175-176
        return wsats, prods

    def sim(self, dt, nTime, wsat0, pbar=True):
        """``model.sim(dt, nTime, wsat0, pbar=False)`` -> ``(nTime+1, Nxy)``, row 0 = ``wsat0``
        (HistoryMatch.py:224-225, 362).  ``self.K`` is ``(2, Nx, Ny)``: the reference always sets Kx = Ky, an anisotropic K runs
        through the same kernels with the y-permeability handed over separately."""
        if np.array_equal(self.K[0], self.K[1]):
            wsats, _ = self.sim_ensemble(self.K[0].reshape(1, -1), np.asarray(wsat0).reshape(1, -1), dt=dt, nTime=nTime,
                                         transformed=True, return_history=True)
            return wsats[0].astype(np.float64, copy=False)
        # anisotropic K = (Kx, Ky): a batch of one through the device-resident plan, which takes the y-permeability separately
        from .forward import ForwardPlan

        plan = ForwardPlan(self, 1, dt, nTime, keep_history=True)
        try:
            plan.set_inputs(self.K[0].reshape(1, -1), np.asarray(wsat0, dtype=float).reshape(1, -1), transformed=True,
                            perms_y=self.K[1].reshape(1, -1))
            plan.run()
            self.last_stats = plan.sync()
            wsats, _, status = plan.outputs()
        finally:
            plan.close()
        if status.any():
            raise _lib.HmError(f"simulation failed (status {int(status[0])})")
        inj_ind, inj, prd_ind, prd = self._wells(nTime)
        cols = lambda r: np.broadcast_to(r, (r.shape[0], nTime)).copy()  # noqa: E731
        self.actual_rates = dict(inj=cols(inj), prd=cols(prd))
        return wsats[0].astype(np.float64, copy=False)
