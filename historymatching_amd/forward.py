"""Ensemble forward model on the GPU: the drop-in for ``forward_model`` / ``utils.apply(comp1, ...)``.

Reference: ``forward_model`` notebooks/HistoryMatch.py:383-387, ``comp1`` :358-364, ``utils.apply``
notebooks/tools/utils.py:155-242 (process-pool map over members).  Here the whole ensemble is ONE device
call; multi-GPU runs shard contiguous member blocks over ranks (``historymatching_amd.dist``).
"""

from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from .ressim import ResSim


def perm_transf(x):
    """Host version of the reference's transform (HistoryMatch.py:137-138); the device applies the same
    formula when ``transformed=False``."""
    return 0.1 + np.exp(5 * x)


class ForwardPlan:
    """Device-resident ensemble forward model (``hm_fwd_*`` in include/hm_abi.h).

    Keeps permeability, saturation history and producer series in HBM so callers (bench.py, ES-MDA,
    iterative smoothers) can chain forward runs and updates without host round trips.
    """

    def __init__(self, model: ResSim, N, dt, nTime, keep_history=True, device=None, ctx=None):
        """``ctx``: an own ``_lib.Context`` (= an own HIP stream on the device) instead of the process-wide one of ``device``:
        plans on different contexts run concurrently (member blocks are independent)."""
        self.model, self.N, self.dt, self.nTime = model, int(N), float(dt), int(nTime)
        self.keep_history = bool(keep_history)
        self.ctx = ctx if ctx is not None else _lib.Context.get(model.device if device is None else device)
        self.lib = self.ctx.lib
        self.ft = np.float64 if model.dtype == 64 else np.float32
        inj_ind, inj, prd_ind, prd = model._wells(self.nTime)
        por = None if model.por is None else _lib.as_c(np.asarray(model.por).reshape(-1), np.float64)
        dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
        h = C.c_void_p()
        _lib.check(self.lib.hm_fwd_create(
            self.ctx.handle, self.N, model.Nx, model.Ny, model.Lx, model.Ly, model.nInj, ip(inj_ind), dp(inj),
            inj.shape[1], model.nPrd, ip(prd_ind), dp(prd), prd.shape[1], self.dt, self.nTime, model.vw, model.vo,
            model.swc, model.sor, dp(por), model.dtype, int(self.keep_history), C.byref(h)), "hm_fwd_create")
        self.h = h

    def close(self):
        if self.h:
            self.lib.hm_fwd_destroy(self.h)
            self.h = None

    __del__ = close

    def set_inputs(self, perms, wsat0s=None, transformed=False, perms_y=None):
        """``perms_y``: the y-permeability per member for an anisotropic K (``model.K[1]`` differing from ``model.K[0]``);
        None = Kx = Ky, the reference's case (set_perm, HistoryMatch.py:160-164)."""
        perms = _lib.as_c(perms, np.float64)
        if perms.shape != (self.N, self.model.Nxy):
            raise ValueError(f"perms must have shape {(self.N, self.model.Nxy)}, got {perms.shape}")
        if wsat0s is not None:
            wsat0s = _lib.as_c(wsat0s, self.ft)
            if wsat0s.shape != perms.shape:
                raise ValueError(f"wsat0s must have shape {perms.shape}, got {wsat0s.shape}")
        _lib.check(self.lib.hm_fwd_set_inputs(self.h, _lib.ptr(perms), int(bool(transformed)), _lib.ptr(wsat0s)),
                   "hm_fwd_set_inputs")
        if perms_y is not None:
            perms_y = _lib.as_c(perms_y, np.float64)
            if perms_y.shape != perms.shape:
                raise ValueError(f"perms_y must have shape {perms.shape}, got {perms_y.shape}")
        _lib.check(self.lib.hm_fwd_set_perm_y(self.h, _lib.ptr(perms_y), int(bool(transformed))), "hm_fwd_set_perm_y")

    def set_inputs_device(self, perm_ptr, perm_dtype=64, transformed=False):
        """Permeability input from a DEVICE buffer of this context (e.g. ``UpdatePlan.device_ptr("E_out")``); initial
        saturation zero.  Asynchronous: ordered on the context's stream behind whatever produced the buffer."""
        _lib.check(self.lib.hm_fwd_set_inputs_device(self.h, C.c_void_p(perm_ptr), int(perm_dtype), int(bool(transformed))),
                   "hm_fwd_set_inputs_device")

    def set_variant(self, pressure=0, saturation=0):
        """0 = fastest applicable kernel, 1 = generic kernels (the in-library correctness baseline)."""
        _lib.check(self.lib.hm_fwd_set_variant(self.h, int(pressure), int(saturation)), "hm_fwd_set_variant")

    def set_debug(self, key, value):
        """Test / experiment knobs (hm_fwd_set_debug): "nd_force_fallback" = member handed to the CG every step (-1 off), "nd_cap" =
        members per block of the larger grids' direct solver (before the first run)."""
        _lib.check(self.lib.hm_fwd_set_debug(self.h, key.encode(), int(value)), "hm_fwd_set_debug")

    def set_solver(self, rtol=1e-12, max_iter=None):
        """Conjugate-gradient pressure solver (always used when Ny > 128; pressure variant 9 elsewhere)."""
        if max_iter is None:
            max_iter = 40 * max(self.model.Nx, self.model.Ny) + 1000
        _lib.check(self.lib.hm_fwd_set_solver(self.h, float(rtol), int(max_iter)), "hm_fwd_set_solver")

    def run(self, first_step=0, n_steps=None):
        n = self.nTime - first_step if n_steps is None else n_steps
        _lib.check(self.lib.hm_fwd_run(self.h, int(first_step), int(n)), "hm_fwd_run")

    def pressure_only(self, k=0):
        _lib.check(self.lib.hm_fwd_pressure_only(self.h, int(k)), "hm_fwd_pressure_only")

    def saturation_only(self, k=0):
        _lib.check(self.lib.hm_fwd_saturation_only(self.h, int(k)), "hm_fwd_saturation_only")

    def sync(self):
        st = _lib.hm_stats()
        _lib.check(self.lib.hm_fwd_sync(self.h, C.byref(st)), "hm_fwd_sync")
        return dict(st.asdict(), nd_fallbacks=int(self.lib.hm_fwd_nd_fallbacks(self.h)), team_retries=int(self.lib.hm_fwd_team_retries(self.h)),
                    slab_redos=int(self.lib.hm_fwd_slab_redos(self.h)))

    def run_to_host(self, out=None):
        """All steps, then ``(wsats, prods, status, stats)`` on the host; a large saturation history is copied out time index
        by time index while the later steps run (hm_fwd_run_to_host).  ``out``: C-contiguous ``(wsats, prods, status)`` to fill
        (slices of a larger ensemble's arrays along the member axis) instead of fresh arrays."""
        m = self.model
        if out is None:
            wsats = np.empty((self.N, self.nTime + 1, m.Nxy) if self.keep_history else (self.N, m.Nxy), dtype=self.ft)
            prods = np.empty((self.N, self.nTime, m.nPrd), dtype=self.ft)
            status = np.zeros(self.N, dtype=np.int32)
        else:
            wsats, prods, status = out
            assert all(a.flags.c_contiguous and len(a) == self.N for a in out) and wsats.dtype == prods.dtype == self.ft
        st = _lib.hm_stats()
        _lib.check(self.lib.hm_fwd_run_to_host(self.h, _lib.ptr(wsats), _lib.ptr(prods), status.ctypes.data_as(C.POINTER(C.c_int)),
                                               C.byref(st)), "hm_fwd_run_to_host")
        return wsats, prods, status, st.asdict()

    def outputs(self, want_wsats=True):
        m = self.model
        wsats = None
        if want_wsats:
            wsats = np.empty((self.N, self.nTime + 1, m.Nxy) if self.keep_history else (self.N, m.Nxy), dtype=self.ft)
        prods = np.empty((self.N, self.nTime, m.nPrd), dtype=self.ft)
        status = np.zeros(self.N, dtype=np.int32)
        _lib.check(self.lib.hm_fwd_get_outputs(self.h, _lib.ptr(wsats), _lib.ptr(prods),
                                               status.ctypes.data_as(C.POINTER(C.c_int))), "hm_fwd_get_outputs")
        return wsats, prods, status

    _shapes = {"P": "c", "K": "c", "S": "c", "Vx": "x", "Vy": "y", "TX": "x", "TY": "y", "nts": "t"}

    def _field_array(self, name):
        m = self.model
        kind = self._shapes[name]
        if kind == "c":
            return np.empty((self.N, m.Nx, m.Ny), dtype=self.ft if name == "S" else np.float64)
        if kind == "x":
            return np.empty((self.N, m.Nx + 1, m.Ny))
        if kind == "y":
            return np.empty((self.N, m.Nx, m.Ny + 1))
        return np.empty((self.N, self.nTime), dtype=np.int32)

    def get_field(self, name):
        out = self._field_array(name)
        _lib.check(self.lib.hm_fwd_get_field(self.h, name.encode(), _lib.ptr(out)), "hm_fwd_get_field")
        return out

    def set_field(self, name, arr):
        ref = self._field_array(name)
        arr = np.ascontiguousarray(arr, dtype=ref.dtype).reshape(ref.shape)
        _lib.check(self.lib.hm_fwd_set_field(self.h, name.encode(), _lib.ptr(arr)), "hm_fwd_set_field")

    def device_ptr(self, name):
        return self.lib.hm_fwd_device_ptr(self.h, name.encode())


def default_blocks(model, N):
    """Member blocks a forward run of ``N`` members is split into (each on its own HIP stream): three (two below 768 members) for a
    large ensemble on the 128 x 128 kernels -- the pressure solve of one block (latency at modest occupancy, matrix pipes) runs beside
    the saturation sweeps (vector pipe) of the others and fills the partial last round of their launches (1000 workgroups are 3.9 rounds
    of 256 CUs) -- one otherwise.  Config 2 on one MI355X (profiles/diag/blocks_time.py, round 5): 565 / 549 / 535-540 / 552-564 / 556 ms
    per pass with 1 / 2 / 3 / 4 / 5 equal blocks.  Members are independent (HistoryMatch.py:376-380): the results are bit-identical
    to the one-block run."""
    if model.Nx != 128 or model.Ny != 128:
        return 1
    return 3 if N >= 768 else 2 if N >= 512 else 1


def block_bounds(N, blocks):
    """Member ranges of ``blocks`` member blocks: equal sizes rounded to a MULTIPLE OF 32 members, the last block takes the rest.  Workgroups
    are dealt round-robin to the 8 XCDs (each with an L2 of its own); the per-member launches of the 128 x 128 kernels index members fastest,
    and ``k_nd_sub`` groups four consecutive members into a workgroup -- with a multiple of 32 members per block every launch of a time step
    puts a member's workgroups on the same XCD, so what one launch writes (coefficient block, update matrices, factor) the next finds in that
    L2.  Measured at config 2 on one MI355X (profiles/r06/blocks_time.txt): three blocks of 320 / 320 / 360 members 77.0-77.4 k
    ensemble-steps/s, 288 / 288 / 424 77.2-77.5 k, equal thirds (333 / 333 / 334) 75.6-75.9 k, 304 / 304 / 392 75.9 k.  Results do not depend on
    the partition (members are independent)."""
    N, blocks = int(N), int(blocks)
    if blocks <= 1:
        return [0, N]
    size = max(32, int(round(N / blocks / 32.0)) * 32)
    if size * (blocks - 1) >= N:
        return [int(b) for b in np.linspace(0, N, blocks + 1).astype(int)]
    return [i * size for i in range(blocks)] + [N]


def merge_block_stats(sts, sizes):
    """Statistics of member blocks that ran side by side as one record: device times are the longest block's, counts add up, the
    ``mean_*`` entries are member-weighted means over the blocks (one division, whatever the number and sizes of the blocks)."""
    st = dict(sts[0])
    total = float(sum(sizes))
    for key, v in sts[0].items():
        if key.startswith("ms_"):
            st[key] = max(r[key] for r in sts)
        elif key.startswith("mean_"):
            st[key] = sum(r[key] * n for r, n in zip(sts, sizes)) / total
        elif isinstance(v, (int, float)):
            st[key] = sum(r[key] for r in sts)
    st["blocks"] = len(sts)
    return st


class BlockedForwardPlan:
    """A device-resident ensemble as ``default_blocks`` member blocks, each a ``ForwardPlan`` on a stream of its own; ``run`` queues
    the blocks' launches interleaved time step by time step from the calling thread (nothing waits for the device).  The same
    arrangement ``make_forward_model`` uses for its host-array calls (there with a host thread per block, for the copies)."""

    def __init__(self, model: ResSim, N, dt, nTime, keep_history=True, device=None, blocks=None, bounds=None):
        device = model.device if device is None else device
        if bounds is None:
            blocks = default_blocks(model, N) if blocks is None else int(blocks)
            bounds = block_bounds(N, blocks)
        self.bounds = [int(b) for b in bounds]
        assert self.bounds[0] == 0 and self.bounds[-1] == N and all(a < b for a, b in zip(self.bounds[:-1], self.bounds[1:]))
        self.N, self.nTime, self.model, self.keep_history = int(N), int(nTime), model, bool(keep_history)
        ctxs = [_lib.Context.get(device)] + [_lib.Context.secondary(device, i) for i in range(len(self.bounds) - 2)]
        self.plans = [ForwardPlan(model, hi - lo, dt, nTime, keep_history=keep_history, ctx=c)
                      for c, lo, hi in zip(ctxs, self.bounds[:-1], self.bounds[1:])]
        self.ft = self.plans[0].ft

    def close(self):
        for pl in self.plans:
            pl.close()

    def set_variant(self, pressure=0, saturation=0):
        for pl in self.plans:
            pl.set_variant(pressure, saturation)

    def set_inputs(self, perms, wsat0s=None, transformed=False):
        perms = _lib.as_c(perms, np.float64)
        if perms.shape != (self.N, self.model.Nxy):
            raise ValueError(f"perms must have shape {(self.N, self.model.Nxy)}, got {perms.shape}")
        for pl, lo, hi in zip(self.plans, self.bounds[:-1], self.bounds[1:]):
            pl.set_inputs(perms[lo:hi], None if wsat0s is None else wsat0s[lo:hi], transformed=transformed)

    def run(self, first_step=0, n_steps=None):
        n = self.nTime - first_step if n_steps is None else n_steps
        if len(self.plans) == 1:
            return self.plans[0].run(first_step, n)
        for k in range(first_step, first_step + n):
            for pl in self.plans:
                pl.run(k, 1)

    def sync(self):
        """Waits for every block; device times are the longest block's, counts add up."""
        sts = [pl.sync() for pl in self.plans]
        return merge_block_stats(sts, [pl.N for pl in self.plans])

    def outputs(self, want_wsats=True):
        outs = [pl.outputs(want_wsats) for pl in self.plans]
        return (np.concatenate([o[0] for o in outs]) if want_wsats else None, np.concatenate([o[1] for o in outs]),
                np.concatenate([o[2] for o in outs]))


def make_forward_model(model: ResSim, dt, nTime, wsat0=None, return_history=True):
    """Build the notebook's ``forward_model`` for a given base ``model`` (HistoryMatch.py:358-387).

    The returned function has the reference signature ``forward_model(*args, leave=True, desc="Ens-run",
    **kwargs)`` -> ``[wsats (N, nTime+1, Nxy), prods (N, nTime, nPrd)]``: the first ensemble argument is the
    pre-permeability ensemble ``(N, Nxy)``, the optional second (or ``wsat0=`` keyword) the per-member
    initial saturation ``(N, Nxy)`` (HistoryMatch.py:1224-1227).  ``leave``/``desc`` only drove the
    reference's progress bar and are accepted and ignored."""
    default_wsat0 = np.zeros(model.Nxy) if wsat0 is None else np.asarray(wsat0, dtype=float)
    cache = {}

    def forward_model(*args, leave=True, desc="Ens-run", **kwargs):
        args = list(args) + list(kwargs.values())  # utils.py:172-173: kwargs become positional ensembles
        if not 1 <= len(args) <= 2:
            raise TypeError(f"comp1() takes 1 or 2 ensemble arguments ({len(args)} given)")
        perms = np.asarray(args[0])
        if len(args) == 2:
            wsat0s = np.asarray(args[1])
            if len(wsat0s) != len(perms):
                raise ValueError("zip() arguments have different lengths")  # zip(strict=True), utils.py:175
        else:
            wsat0s = np.broadcast_to(default_wsat0, perms.shape)
        perms = _lib.as_c(perms, np.float64)
        if perms.ndim != 2 or perms.shape[1] != model.Nxy:
            raise ValueError(f"perms must have shape (N, {model.Nxy}), got {perms.shape}")
        if np.shape(wsat0s) != perms.shape:
            raise ValueError(f"wsat0s must have shape {perms.shape}, got {np.shape(wsat0s)}")
        # The device plans (22 GB of buffers at N_e = 1000, 128 x 128) are kept between calls: creating and freeing them costs
        # 0.25 s per call, a fifth of the run itself.  They are rebuilt when the ensemble size or the model's wells / rates /
        # fluid / porosity change.
        # A large ensemble on the 128 x 128 kernels runs as member blocks on streams of their own (default_blocks), each driven by
        # its own host thread (the copies of a block's history go out beside the other blocks' kernels); bit-identical to one block.
        N = len(perms)
        blocks = default_blocks(model, N)
        inj_ind, inj, prd_ind, prd = model._wells(nTime)
        sig = (N, blocks, model.dtype, inj_ind.tobytes(), inj.tobytes(), prd_ind.tobytes(), prd.tobytes(), model.vw, model.vo, model.swc,
               model.sor, None if model.por is None else np.asarray(model.por, dtype=float).tobytes())
        if cache.get("sig") != sig:
            release()
            bounds = block_bounds(N, blocks)
            # the further blocks' own streams: one secondary context per device and block index, shared process-wide
            ctxs = [None] + [_lib.Context.secondary(model.device, i) for i in range(blocks - 1)]
            cache["plans"] = [(ForwardPlan(model, hi - lo, dt, nTime, keep_history=return_history, ctx=c), lo, hi)
                              for c, lo, hi in zip(ctxs, bounds[:-1], bounds[1:])]
            cache["sig"] = sig
        ft = np.float64 if model.dtype == 64 else np.float32
        wsats = np.empty((N, nTime + 1, model.Nxy) if return_history else (N, model.Nxy), dtype=ft)
        prods = np.empty((N, nTime, model.nPrd), dtype=ft)
        status = np.zeros(N, dtype=np.int32)
        results = [None] * blocks

        def run_block(b):
            plan, lo, hi = cache["plans"][b]
            try:
                plan.set_inputs(perms[lo:hi], wsat0s[lo:hi], transformed=False)
                results[b] = plan.run_to_host(out=(wsats[lo:hi], prods[lo:hi], status[lo:hi]))[3]
            except BaseException as e:  # re-raised by the caller's thread
                results[b] = e

        workers = [threading.Thread(target=run_block, args=(b,)) for b in range(1, blocks)]
        for t in workers:
            t.start()
        run_block(0)
        for t in workers:
            t.join()
        for r in results:
            if isinstance(r, BaseException):
                raise r
        # the blocks run side by side: device times are the longest block's, counts add up, means are member-weighted
        model.last_stats = merge_block_stats(results, [hi - lo for _, lo, hi in cache["plans"][:blocks]])
        if status.any():
            bad = np.flatnonzero(status)
            raise _lib.HmError(f"forward model failed for members {bad[:8].tolist()} (status {status[bad[:8]].tolist()}): "
                               "1=non-positive pivot in pressure solve, 2=bad CFL, 4=non-finite saturation, "
                               "8=CG pressure solver did not converge")
        cols = lambda r: np.broadcast_to(r, (r.shape[0], nTime)).copy()  # noqa: E731
        model.actual_rates = dict(inj=cols(inj), prd=cols(prd))  # Optimise.py:175-176
        return [wsats, prods]

    def comp1(perm, wsat0=None):
        """The reference's per-member composite (HistoryMatch.py:358-364): one pre-permeability field ``(Nxy,)`` ->
        ``(wsats (nTime+1, Nxy), prods (nTime, nPrd))``; a batch of one through the same device path."""
        w0 = default_wsat0 if wsat0 is None else np.asarray(wsat0, dtype=float)
        wsats, prods = forward_model(np.asarray(perm)[None, :], w0[None, :])
        return wsats[0], prods[0]

    def comp1_batched(perms, wsat0=None):
        """``utils.apply(comp1, perms[, wsat0s])`` in one device call: the list of per-member ``(wsats, prods)`` pairs the
        notebook's ``forward_model`` transposes (HistoryMatch.py:383-387)."""
        wsats, prods = forward_model(perms) if wsat0 is None else forward_model(perms, wsat0)
        return list(zip(wsats, prods))

    comp1.batched = comp1_batched

    def release():
        """Free the cached device plans (they are rebuilt by the next call)."""
        for plan, _, _ in cache.get("plans", []):
            plan.close()
        cache["plans"], cache["sig"] = [], None

    forward_model.comp1 = comp1
    forward_model.release = release
    return forward_model
