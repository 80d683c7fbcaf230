"""Multi-GPU layer: one process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

Sharding follows the reference's only parallel axis -- independent ensemble members (``utils.apply`` process
pool, notebooks/tools/utils.py:201-224; "embarrassingly parallelizable" notebooks/HistoryMatch.py:376-380):
  * forward model: contiguous member blocks per rank, NO data-path collective;
  * update: rows of E stay on their rank; two reduction points per update (column sums: M+n_obs values; the
    Gram pair S^T S, X^T S: n_obs*(n_obs+M) values), SURVEY.md 8e.  Everything else is row-local.
torch is used only for rendezvous and the collective; compute stays behind the C ABI.
"""

from __future__ import annotations

import os

import numpy as np


def shard_bounds(N, world_size, rank):
    """Contiguous block [lo, hi) of `rank`; the first N % world_size ranks hold one extra member."""
    base, extra = divmod(int(N), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Comm:
    """Thin wrapper over an initialised torch.distributed process group (or a single process)."""

    def __init__(self, group=None):
        self.group = group
        try:
            import torch.distributed as td

            self.td = td if td.is_available() and td.is_initialized() else None
        except Exception:  # torch absent: single process only
            self.td = None
        self.rank = self.td.get_rank(group) if self.td else 0
        self.world_size = self.td.get_world_size(group) if self.td else 1

    def all_reduce_sum(self, arr):
        """Sum a host array over ranks (host-staged: works for gloo and, through a device tensor, for nccl)."""
        if not self.td or self.world_size == 1:
            return arr
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.td.get_backend(self.group) == "nccl":
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            t = t.to(dev)
            self.td.all_reduce(t, group=self.group)
            return t.cpu().numpy()
        self.td.all_reduce(t, group=self.group)
        return t.numpy()

    @property
    def backend(self):
        return self.td.get_backend(self.group) if self.td else None

    def all_reduce_device(self, ptr, n, dtype, force=False):
        """In-place sum over ranks of ``n`` elements of ``dtype`` at DEVICE address ``ptr`` (a buffer owned by the C-ABI
        library) with RCCL: the buffer is wrapped zero-copy as a torch tensor through ``__cuda_array_interface__``;
        no host staging.  The caller has synchronised the library's stream (``plan.sync()``); this call returns
        after the collective has completed on the device."""
        if not self.td or (self.world_size == 1 and not force):
            return
        import torch

        class _Dev:  # minimal CUDA-array-interface carrier
            pass

        d = _Dev()
        d.__cuda_array_interface__ = {"shape": (int(n),), "typestr": np.dtype(dtype).str, "data": (int(ptr), False), "version": 2}
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        t = torch.as_tensor(d, device=dev)
        self.td.all_reduce(t, group=self.group)
        torch.cuda.synchronize(dev)

    def all_gather_rows(self, arr):
        """Concatenate per-rank row blocks (possibly ragged) in rank order."""
        if not self.td or self.world_size == 1:
            return arr
        out = [None] * self.world_size
        self.td.all_gather_object(out, np.ascontiguousarray(arr), group=self.group)
        return np.concatenate(out, axis=0)

    def barrier(self):
        if self.td and self.world_size > 1:
            self.td.barrier(group=self.group)


def forward_model_sharded(local_forward, perms, wsat0s=None, comm=None, gather=True):
    """Run ``local_forward(perms_block[, wsat0s_block]) -> [wsats, prods]`` on this rank's member block.
    With ``gather`` the full ``[wsats, prods]`` (member order preserved, HistoryMatch.py:387) is returned on
    every rank; otherwise only the local block."""
    comm = comm or Comm()
    lo, hi = shard_bounds(len(perms), comm.world_size, comm.rank)
    args = [np.asarray(perms)[lo:hi]]
    if wsat0s is not None:
        if len(wsat0s) != len(perms):
            raise ValueError("ensemble arguments have different lengths")  # zip(strict=True), utils.py:175
        args.append(np.asarray(wsat0s)[lo:hi])
    wsats, prods = local_forward(*args)
    if gather:
        return [comm.all_gather_rows(wsats), comm.all_gather_rows(prods)]
    return [wsats, prods]


def sharded_update(plan, comm=None, fetch=True):
    """Drive a row-sharded update plan (``update.UpdatePlan`` or any object with the same
    ``phase / get_reduce / set_reduce / sync / output`` methods) through its three phases, summing the two
    reduce buffers over ranks in between.  Returns this rank's rows of the updated ensemble (``fetch=False``: leaves
    them on the device and returns the plan's statistics instead)."""
    comm = comm or Comm()
    device_direct = comm.backend == "nccl" and hasattr(plan, "reduce_buffer")
    for ph in range(3):
        plan.phase(ph)
        if ph < 2 and comm.world_size > 1:
            if device_direct:  # RCCL on the library's own device buffers (xGMI), no PCIe round trip
                plan.sync()
                for which in plan.REDUCE_AFTER_PHASE[ph]:
                    ptr, n, dt = plan.reduce_buffer(which)
                    comm.all_reduce_device(ptr, n, dt)
            else:  # gloo / CPU test doubles: host-staged
                for which in plan.REDUCE_AFTER_PHASE[ph]:
                    plan.set_reduce(which, comm.all_reduce_sum(plan.get_reduce(which)))
    st = plan.sync()
    return plan.output() if fetch else st


def es_mda_sharded(model, prior_local, obs, R12, dt, nTime, n_iter=4, seed=0, comm=None, dtype=32, taper=None, device=None,
                   stats=None):
    """ES-MDA over ranks with every rank's members resident in its GPU's HBM for the whole assimilation (BASELINE configs 4
    and 5; the single-GPU form is ``update.es_mda_device``).  ``prior_local``: this rank's contiguous block of the
    ``(N_total, M)`` prior (``shard_bounds``).  Per pass: forward model of the local members (no communication) -> their
    simulated observations handed to the update plan on the device -> the three phases of the row-sharded analysis step with
    its two all-reduces (RCCL on the library's buffers with the nccl backend, host-staged with gloo) -> pointer swap.
    ``taper``: ``(M, n_obs)`` localisation coefficients for the localised analysis (``ens_update0_loc``,
    HistoryMatch.py:774-797); None = global.  The observation perturbations are rows of ONE ``(N_total, n_obs)`` normal
    matrix drawn identically on every rank from ``seed``, so the result does not depend on the number of ranks.
    Returns this rank's rows of the posterior."""
    import ctypes as C

    import scipy.linalg as sla

    from . import _lib
    from .forward import ForwardPlan
    from .update import UpdatePlan

    comm = comm or Comm()
    E0 = np.asarray(prior_local)
    Nl, M = E0.shape
    counts = comm.all_gather_rows(np.array([[Nl]]))[:, 0] if comm.world_size > 1 else np.array([Nl])
    N, lo = int(counts.sum()), int(counts[: comm.rank].sum())
    n_obs = len(obs)
    if n_obs != nTime * model.nPrd:
        raise ValueError(f"len(obs) = {n_obs} != nTime * nPrd = {nTime * model.nPrd}")
    rng = np.random.RandomState(seed)
    fwd = ForwardPlan(model, Nl, dt, nTime, keep_history=False, device=device)
    upd = UpdatePlan(N, Nl, M, n_obs, dtype=dtype, localized=taper is not None, device=device)
    alpha = float(n_iter)
    upd.set_inputs(E=E0, obs=obs, decorr=sla.inv(R12.T) / np.sqrt(alpha), taper=taper)
    ms_fwd = ms_upd = 0.0
    try:
        for _ in range(n_iter):
            fwd.set_inputs_device(upd.device_ptr("E"), dtype, transformed=False)
            fwd.run()
            upd.set_inputs_device(obs_ens_ptr=fwd.device_ptr("prods"), obs_dtype=model.dtype)
            upd.set_inputs(perturbs=np.sqrt(alpha) * (rng.randn(N, n_obs)[lo:lo + Nl] @ R12.T))
            ms_fwd += fwd.sync()["ms_total"]
            _, _, status = fwd.outputs(want_wsats=False)
            bad = comm.all_reduce_sum(np.array([float(status.any())]))[0] if comm.world_size > 1 else float(status.any())
            if bad:  # every rank leaves together
                raise _lib.HmError(f"forward model failed on {int(bad)} rank(s); here for members {np.flatnonzero(status)[:8].tolist()}")
            st = sharded_update(upd, comm, fetch=False) if comm.world_size > 1 else upd.run_local()
            ms_upd += st["ms_update"]
            upd.swap()
        out = np.empty((Nl, M), dtype=upd.ft)
        _lib.check(upd.lib.hm_copy_to_host(upd.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(upd.device_ptr("E")), out.nbytes),
                   "hm_copy_to_host")
    finally:
        fwd.close()
        upd.close()
    if stats is not None:
        stats.update(ms_forward=ms_fwd, ms_update=ms_upd)
    return out.astype(float)
