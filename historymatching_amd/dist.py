"""Multi-GPU layer: one process per GPU, no PyTorch.

Sharding follows the reference's only parallel axis -- independent ensemble members (``utils.apply`` process
pool, notebooks/tools/utils.py:201-224; "embarrassingly parallelizable" notebooks/HistoryMatch.py:376-380):
  * forward model: contiguous member blocks per rank, NO data-path collective;
  * update: rows of E stay on their rank; two reduction points per update (column sums: M+n_obs values; the
    Gram pair S^T S, X^T S: n_obs*(n_obs+M) values); the localised analysis additionally shards its
    per-element solves by state column and all-gathers the weights.  Everything else is row-local.

Two channels per rank:
  * ``HostChannel`` -- a localhost socket star (rank 0 is the hub) for rendezvous, small host objects, status flags,
    barriers and ordered gathers; sums are formed in rank order, so results do not depend on arrival order;
  * RCCL (``Comm.enable_rccl``) -- the library's own communicator (``hm_comm_*`` in include/hm_abi.h): the update's
    reductions run in place on the plan's device buffers, queued on the context's stream between the phases.
Ranks that share one GPU (tests on a one-GPU box) or have no GPU (CPU tests) use the host channel for the same reductions:
RCCL refuses two ranks of a communicator on one device.

Launch: ``python -m torch.distributed.run --nproc-per-node N script.py`` (or any launcher that sets RANK, WORLD_SIZE,
LOCAL_RANK); ``Comm.from_env()`` does the rest.  The launcher's own MASTER_PORT store is not used.
"""

from __future__ import annotations

import ctypes as C
import os
import secrets
import time
from multiprocessing.connection import Client, Listener

import numpy as np


def shard_bounds(N, world_size, rank):
    """Contiguous block [lo, hi) of `rank`; the first N % world_size ranks hold one extra member."""
    base, extra = divmod(int(N), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _remove_own_file(path):
    """Unlink `path` if it exists; a file another user planted there (sticky /tmp) is reported, not tripped over."""
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    except PermissionError as e:
        raise PermissionError(f"rendezvous path {path} exists and belongs to another user; set HM_AMD_RDZV to a private path") from e


def default_rdzv_dir():
    """A directory only this user can enter: $XDG_RUNTIME_DIR when it is one, else /tmp/hm_amd_<uid> created 0700 (and
    checked: a directory someone else made under that name is refused)."""
    xdg = os.environ.get("XDG_RUNTIME_DIR")
    for d in ([xdg] if xdg else []) + [f"/tmp/hm_amd_{os.geteuid()}"]:
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.stat(d)
            if st.st_uid == os.geteuid() and not (st.st_mode & 0o077):
                return d
        except OSError:
            continue
    raise PermissionError("no private directory for the rendezvous file: set HM_AMD_RDZV")


class HostChannel:
    """Star of localhost connections (``multiprocessing.connection``: length-prefixed messages, HMAC handshake).
    Rank 0 listens on a port the OS picks and publishes ``port key`` in the rendezvous file; the others poll the file and
    connect.  A peer that dies closes its connection, which surfaces as EOFError on the other side: no silent hang."""

    def __init__(self, rank, world_size, rdzv_file, timeout=120.0):
        self.rank, self.world_size = int(rank), int(world_size)
        self.peers = {}
        if self.world_size == 1:
            return
        env_key = os.environ.get("HM_AMD_RDZV_KEY")  # a launcher that owns the environment passes the key there: the file then holds the port only
        if self.rank == 0:
            key = bytes.fromhex(env_key) if env_key else secrets.token_bytes(16)
            _remove_own_file(rdzv_file)  # left behind by a run that crashed
            listener = Listener(("127.0.0.1", 0), authkey=key)
            tmp = f"{rdzv_file}.{os.getpid()}"
            # created exclusively and readable by the owner only: the key (when it is in the file) authenticates a channel that
            # unpickles what it receives
            fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
            with os.fdopen(fd, "w") as fh:
                fh.write(f"{listener.address[1]} {'-' if env_key else key.hex()}\n")
            os.replace(tmp, rdzv_file)
            try:
                try:  # bound the wait for peers that never show up (private attribute of the stdlib Listener: best effort)
                    listener._listener._socket.settimeout(timeout)
                except AttributeError:
                    pass
                while len(self.peers) < self.world_size - 1:
                    conn = listener.accept()
                    self.peers[int(conn.recv())] = conn
            finally:
                listener.close()
                _remove_own_file(rdzv_file)
        else:
            t_end = time.time() + timeout
            while True:
                try:
                    st = os.stat(rdzv_file)
                    if st.st_uid != os.geteuid() or (st.st_mode & 0o077):
                        raise PermissionError(f"{rdzv_file} is not a private file of this user")
                    with open(rdzv_file) as fh:
                        port, key = fh.read().split()
                    conn = Client(("127.0.0.1", int(port)), authkey=bytes.fromhex(env_key if key == "-" else key))
                    break
                except Exception as e:  # no file yet, a stale file (refused / wrong key), a file that is not ours: try again
                    if time.time() > t_end:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous through {rdzv_file} within {timeout} s ({type(e).__name__}: {e})")
                    time.sleep(0.05)
            conn.send(self.rank)
            self.peers[0] = conn

    def gather(self, obj):
        """Rank 0 returns the ranks' objects in rank order; the others None."""
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [self.peers[r].recv() for r in range(1, self.world_size)]
        self.peers[0].send(obj)
        return None

    def bcast(self, obj):
        """Rank 0's object on every rank."""
        if self.world_size == 1:
            return obj
        if self.rank == 0:
            for r in range(1, self.world_size):
                self.peers[r].send(obj)
            return obj
        return self.peers[0].recv()

    def all_gather(self, obj):
        return self.bcast(self.gather(obj))

    def close(self):
        for c in self.peers.values():
            c.close()
        self.peers = {}


class Comm:
    """The ranks of one job: host channel always, RCCL communicator once ``enable_rccl`` has succeeded."""

    def __init__(self, rank=0, world_size=1, rdzv_file=None, local_rank=None, timeout=120.0):
        self.rank, self.world_size = int(rank), int(world_size)
        self.local_rank = self.rank if local_rank is None else int(local_rank)
        if self.world_size > 1 and rdzv_file is None:
            raise ValueError("a rendezvous file path is needed when world_size > 1")
        self.host = HostChannel(self.rank, self.world_size, rdzv_file, timeout)
        self.rccl = None        # hm_comm* handle
        self.ctx = None         # the context the RCCL communicator is bound to (device + stream)
        self.rccl_error = None  # why device collectives are not in use (text), if they were asked for

    @classmethod
    def from_env(cls, timeout=120.0):
        """RANK / WORLD_SIZE / LOCAL_RANK as set by ``torch.distributed.run``, ``mpirun`` wrappers and the like.  The
        rendezvous file is ``$HM_AMD_RDZV`` or a name built from the launcher's pid and MASTER_PORT (both are the same for all
        ranks of a job and differ between concurrent jobs) inside a directory only this user can enter (``default_rdzv_dir``);
        ``$HM_AMD_RDZV_KEY`` (hex), when the launcher sets it, keeps the channel's key out of the file."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        rdzv = os.environ.get("HM_AMD_RDZV")
        if not rdzv and world > 1:
            rdzv = os.path.join(default_rdzv_dir(), f"rdzv_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL's dmabuf IPC between the ranks' processes
        return cls(rank, world, rdzv, local_rank=local, timeout=timeout)

    # ---------------------------------------------------------------- host-side collectives (NumPy arrays)
    def all_reduce_sum(self, arr):
        """Sum a host array over ranks, formed in rank order on rank 0 (bit-reproducible)."""
        if self.world_size == 1:
            return arr
        parts = self.host.gather(np.ascontiguousarray(arr))
        total = None
        if parts is not None:
            total = parts[0].copy()
            for p in parts[1:]:
                total += p
        return self.host.bcast(total)

    def all_reduce_max(self, value):
        if self.world_size == 1:
            return value
        return max(self.host.all_gather(value))

    def all_gather_rows(self, arr):
        """Concatenate per-rank row blocks (possibly ragged) in rank order."""
        if self.world_size == 1:
            return arr
        return np.concatenate(self.host.all_gather(np.ascontiguousarray(arr)), axis=0)

    def barrier(self):
        if self.world_size > 1:
            self.host.all_gather(None)

    def raise_if_any(self, error, what="a rank failed"):
        """Every rank calls this with its own exception (or None) before the next collective: if any rank failed, all
        raise together instead of leaving the healthy ranks blocked in a collective the failed one never enters."""
        if self.world_size == 1:
            if error is not None:
                raise error
            return
        msgs = self.host.all_gather(None if error is None else f"rank {self.rank}: {type(error).__name__}: {error}")
        bad = [m for m in msgs if m]
        if error is not None:
            raise error
        if bad:
            from ._lib import HmError

            raise HmError(f"{what}: " + "; ".join(bad))

    # ---------------------------------------------------------------- device-side collectives (RCCL through the C ABI)
    def enable_rccl(self, ctx=None, force_single=False):
        """Create the library's RCCL communicator on ``ctx`` (default: the context of device LOCAL_RANK).  Returns True when
        device collectives are available afterwards.  All ranks call it.  With one rank nothing is created unless
        ``force_single`` (tests: the RCCL path on one GPU).  A failure on any rank (e.g. two ranks on one GPU: "Duplicate
        GPU detected") leaves every rank on the host channel, the reason in ``rccl_error``."""
        from . import _lib

        if self.rccl is not None:
            return True
        if self.world_size == 1 and not force_single:
            return False
        # Phase 1, every rank: a device context, librccl opened and its symbols bound (rank 0: hm_comm_unique_id, whose id is the one
        # used; the others: hm_comm_probe, which starts no bootstrap listener).  The outcome is agreed on BEFORE anyone enters
        # ncclCommInitRank: a rank that cannot take part (no device, no librccl, a bad LOCAL_RANK) would otherwise leave the healthy
        # ranks blocked in it for ever.
        err, uid, lib = None, None, None
        try:
            ctx = ctx or _lib.Context.get(self.local_rank)
            lib = ctx.lib
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                _lib.check(lib.hm_comm_unique_id(buf), "hm_comm_unique_id")
                uid = buf.raw
            else:
                _lib.check(lib.hm_comm_probe(), "hm_comm_probe")
        except Exception as e:  # no GPU / no RCCL on this rank
            err = e
        if self.world_size > 1:
            pre = [m for m in self.host.all_gather(None if err is None else f"rank {self.rank}: {err}") if m]
            if pre:
                self.rccl_error = "; ".join(pre)
                return False
        elif err is not None:
            self.rccl_error = str(err)
            return False
        # Phase 2: every rank is known to be able to call it
        uid = self.host.bcast(uid)
        h = C.c_void_p()
        try:
            _lib.check(lib.hm_comm_create(ctx.handle, self.rank, self.world_size, uid, C.byref(h)), "hm_comm_create")
        except Exception as e:
            err = e
        msgs = self.host.all_gather(None if err is None else f"rank {self.rank}: {err}") if self.world_size > 1 else [None if err is None else str(err)]
        bad = [m for m in msgs if m]
        if bad:
            if h:
                lib.hm_comm_destroy(h)
            self.rccl_error = "; ".join(bad)
            return False
        self.rccl, self.ctx = h, ctx
        return True

    def device_all_reduce(self, ptr, n, dtype, op="sum"):
        """In-place all-reduce of ``n`` elements (np.float64 / np.float32 / np.int32) at DEVICE address ``ptr`` of the
        communicator's context; asynchronous on its stream."""
        from . import _lib

        code = {np.dtype(np.float64): 64, np.dtype(np.float32): 32, np.dtype(np.int32): 1}[np.dtype(dtype)]
        _lib.check(self.ctx.lib.hm_comm_all_reduce(self.rccl, C.c_void_p(ptr), int(n), code, 0 if op == "sum" else 1), "hm_comm_all_reduce")

    def device_sync(self):
        from . import _lib

        _lib.check(self.ctx.lib.hm_comm_sync(self.rccl), "hm_comm_sync")

    def close(self):
        if self.rccl is not None:
            self.ctx.lib.hm_comm_destroy(self.rccl)
            self.rccl = None
        self.host.close()


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
    """One analysis step of a row-sharded update plan (``update.UpdatePlan`` or any object with the same
    ``phase / get_reduce / set_reduce / sync / output`` methods) over the ranks of ``comm``.

    With an RCCL communicator bound to the plan's context the whole step is one library call (``hm_upd_run_comm``: phases and
    collectives in stream order, no host synchronisation).  Otherwise the same reductions go through the host channel:
    sum of the reduce buffers after phases 0 and 1, and for a localised plan the per-element solves sharded by state
    column (``set_column_shard``), the weights all-gathered (buffer 4) before the row-local apply (phase 3).
    Returns this rank's rows of the updated ensemble (``fetch=False``: leaves them on the device and returns the plan's
    statistics instead)."""
    comm = comm or Comm()
    if comm.rccl is not None and hasattr(plan, "run_comm"):
        if plan.ctx is not comm.ctx:
            raise ValueError(f"the plan lives on device {plan.ctx.device}, the RCCL communicator on device {comm.ctx.device}")
        plan.run_comm(comm.rccl)
    else:
        world, rank = comm.world_size, comm.rank
        shard_cols = world > 1 and getattr(plan, "localized", False) and hasattr(plan, "set_column_shard")
        if shard_cols:
            plan.set_column_shard(rank, world)
        for ph in range(3):
            plan.phase(ph)
            if world == 1:
                continue
            if ph < 2:
                for which in plan.REDUCE_AFTER_PHASE[ph]:
                    plan.set_reduce(which, comm.all_reduce_sum(plan.get_reduce(which)))
            elif shard_cols:
                blocks = plan.get_reduce(4).reshape(world, -1)
                plan.set_reduce(4, comm.all_gather_rows(blocks[rank][None]))
        if shard_cols:
            plan.phase(3)
    st = plan.sync()
    return plan.output() if fetch else st


def es_mda_sharded(model, prior_local, obs, R12, dt, nTime, n_iter=4, seed=0, comm=None, dtype=32, taper=None, device=None,
                   stats=None):
    """ES-MDA over ranks with every rank's members resident in its GPU's HBM for the whole assimilation (BASELINE configs 4
    and 5; the single-GPU form is ``update.es_mda_device``).  ``prior_local``: this rank's contiguous block of the
    ``(N_total, M)`` prior (``shard_bounds``).  Per pass: forward model of the local members (no communication) -> their
    simulated observations handed to the update plan on the device -> the analysis step over the ranks (``sharded_update``:
    RCCL on the plan's device buffers, or the host channel) -> pointer swap.
    ``taper``: ``(M, n_obs)`` localisation coefficients for the localised analysis (``ens_update0_loc``,
    HistoryMatch.py:774-797); None = global.  The observation perturbations are rows of ONE ``(N_total, n_obs)`` normal
    matrix drawn identically on every rank from ``seed``, so the result does not depend on the number of ranks.
    A failure on one rank (forward-model status, HIP error) is agreed on by all ranks before the next collective.
    Returns this rank's rows of the posterior."""
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
    if device is None and comm.ctx is not None:
        device = comm.ctx.device
    fwd = upd = None
    ms_fwd = ms_upd = ms_comm = 0.0
    nd_fallbacks = team_retries = slab_redos = 0
    alpha = float(n_iter)
    try:
        err = None
        try:
            fwd = ForwardPlan(model, Nl, dt, nTime, keep_history=False, device=device)
            upd = UpdatePlan(N, Nl, M, n_obs, dtype=dtype, localized=taper is not None, device=device)
            upd.set_inputs(E=E0, obs=obs, decorr=sla.inv(R12.T) / np.sqrt(alpha), taper=taper)
        except Exception as e:
            err = e
        comm.raise_if_any(err, "plan creation failed")
        for _ in range(n_iter):
            err = None
            perturbs = np.sqrt(alpha) * (rng.randn(N, n_obs)[lo:lo + Nl] @ R12.T)  # drawn on every rank, failed or not
            try:
                upd.set_inputs(perturbs=perturbs)  # (before the pass: nothing of the update waits for the host afterwards)
                fwd.set_inputs_device(upd.device_ptr("E"), dtype, transformed=False)
                fwd.run()
                upd.set_inputs_device(obs_ens_ptr=fwd.device_ptr("prods"), obs_dtype=model.dtype)
                st_f = fwd.sync()
                ms_fwd += st_f["ms_total"]
                nd_fallbacks, team_retries, slab_redos = st_f["nd_fallbacks"], st_f["team_retries"], st_f["slab_redos"]  # (cumulative over the plan's life)
                _, _, status = fwd.outputs(want_wsats=False)
                if status.any():
                    raise _lib.HmError(f"forward model failed for members {(lo + np.flatnonzero(status))[:8].tolist()} "
                                       f"(status {status[np.flatnonzero(status)[:8]].tolist()})")
            except Exception as e:
                err = e
            comm.raise_if_any(err, "forward model failed")
            st = sharded_update(upd, comm, fetch=False) if comm.world_size > 1 else upd.run_local()
            ms_upd += st["ms_update"]
            ms_comm += st.get("ms_comm", 0.0)
            upd.swap()
        out = np.empty((Nl, M), dtype=upd.ft)
        _lib.check(upd.lib.hm_copy_to_host(upd.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(upd.device_ptr("E")), out.nbytes),
                   "hm_copy_to_host")
    finally:
        if fwd is not None:
            fwd.close()
        if upd is not None:
            upd.close()
    if stats is not None:
        stats.update(ms_forward=ms_fwd, ms_update=ms_upd, ms_comm=ms_comm, nd_fallbacks=nd_fallbacks, team_retries=team_retries, slab_redos=slab_redos)
    return out.astype(float)
