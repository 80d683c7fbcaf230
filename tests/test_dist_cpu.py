"""N>1 path on CPU: world_size-2 runs drive the SAME host logic the GPU ranks use (historymatching_amd.dist: member
sharding, the reduction points of the sharded update incl. the column-sharded localised solves, ordered gather, failure
agreement) -- once over the product's own host channel (localhost sockets, no PyTorch) and once over a `gloo` process group
through a test-side adapter with the same methods.  The per-rank compute is a NumPy test double with the device plan's
interface (phases / reduce buffers), built from the oracle's formulas -- tests may use the oracle, the product path never does."""
import os
import socket
import tempfile

import numpy as np
import pytest

from historymatching_amd.dist import Comm, forward_model_sharded, shard_bounds, sharded_update
from oracle import es


class HostShardPlan:
    """Same phase structure as update.UpdatePlan / csrc/update.hip, in NumPy."""

    REDUCE_AFTER_PHASE = {0: (0, 1), 1: (2, 3)}

    def __init__(self, N_total, E, obs_ens, obs, perturbs, decorr, taper=None, cutoff=1e-2):
        self.N, self.E, self.obs_ens, self.obs, self.perturbs, self.decorr = N_total, E, obs_ens, obs, perturbs, decorr
        self.taper, self.cutoff = taper, cutoff
        self.localized = taper is not None
        self.red = {}
        self.col = (0, 1)  # (rank, world) of the column shard of the localised solves

    def set_column_shard(self, rank, world):
        self.col = (rank, world)

    def phase(self, k):
        N = self.N
        if k == 0:
            self.red[0], self.red[1] = self.E.sum(0), self.obs_ens.sum(0)
        elif k == 1:
            X = self.E - self.red[0] / N
            Y = self.obs_ens - self.red[1] / N
            self.S = Y @ self.decorr
            self.D = (self.obs - self.obs_ens - self.perturbs) @ self.decorr
            self.red[2], self.red[3] = (X.T @ self.S).ravel(), (self.S.T @ self.S).ravel()
        elif k == 2:
            n_obs = len(self.obs)
            G = self.red[3].reshape(n_obs, n_obs)
            Gxt = self.red[2].reshape(-1, n_obs)
            if self.taper is None:
                self.out = self.E + (self.D @ np.linalg.inv(G + (N - 1) * np.eye(n_obs))) @ Gxt.T
                return
            M = Gxt.shape[0]
            rank, world = self.col
            chunk = -(-M // world)
            Wt = np.zeros((world * chunk, n_obs))  # padded to equal blocks like the device buffer
            for i in range(rank * chunk, min(M, (rank + 1) * chunk)):
                c = np.sqrt(self.taper[i])
                jj = c > self.cutoff
                if jj.any():
                    Ci = np.outer(c[jj], c[jj]) * G[np.ix_(jj, jj)] + (N - 1) * np.eye(jj.sum())
                    Wt[i, jj] = c[jj] * np.linalg.solve(Ci, c[jj] * Gxt[i, jj])
            self.red[4] = Wt.ravel()
            if world == 1:
                self.phase(3)
        else:
            n_obs = len(self.obs)
            Wt = self.red[4].reshape(-1, n_obs)[: self.E.shape[1]]
            self.out = self.E + self.D @ Wt.T

    def get_reduce(self, which):
        return self.red[which]

    def set_reduce(self, which, arr):
        self.red[which] = np.asarray(arr)

    def sync(self):
        return {}

    def output(self):
        return self.out


def test_shard_bounds_cover_and_balance():
    for N, W in [(1000, 8), (7, 3), (4096, 8), (5, 8)]:
        b = [shard_bounds(N, W, r) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == N
        assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    rng = np.random.RandomState(3)
    N, M, n_obs = 23, 60, 16
    E = rng.randn(N, M) + 2.0
    obs_ens = E @ (rng.randn(M, n_obs) / 8)
    _, R12, decorr = es.obs_error_model(4, 4)
    obs = obs_ens[1] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.4)
    return N, E, obs_ens, obs, perturbs, decorr, taper


class GlooComm:
    """Test-side adapter: the methods `dist.sharded_update` / `forward_model_sharded` use, over a torch.distributed gloo group."""

    rccl = None

    def __init__(self):
        import torch.distributed as td

        self.td = td
        self.rank, self.world_size = td.get_rank(), td.get_world_size()

    def all_reduce_sum(self, arr):
        import torch

        t = torch.from_numpy(np.array(arr, copy=True))
        self.td.all_reduce(t)
        return t.numpy()

    def all_gather_rows(self, arr):
        out = [None] * self.world_size
        self.td.all_gather_object(out, np.ascontiguousarray(arr))
        return np.concatenate(out, axis=0)


def _drive(comm, rank, world):
    assert (comm.rank, comm.world_size) == (rank, world)
    N, E, obs_ens, obs, perturbs, decorr, taper = _inputs()
    lo, hi = shard_bounds(N, world, rank)
    res = {}
    for name, tp in (("global", None), ("local", taper)):
        plan = HostShardPlan(N, E[lo:hi], obs_ens[lo:hi], obs, perturbs[lo:hi], decorr, taper=tp)
        res[name] = comm.all_gather_rows(sharded_update(plan, comm))
        if tp is not None:
            assert plan.col == (rank, world)  # the localised solves were column-sharded

    # forward model: member blocks, ordered gather, per-member wsat0 zipped along
    def local_forward(perms, wsat0s):
        return [perms[:, None, :] * 2 + wsat0s[:, None, :], perms[:, :3, None] + np.zeros((1, 1, 4))]

    perms = np.arange(N * 5, dtype=float).reshape(N, 5)
    w, p = forward_model_sharded(local_forward, perms, perms * 0.5, comm=comm)
    res["w"], res["p"] = w, p
    return res


def _worker_sockets(rank, world, rdzv, q):
    comm = Comm(rank, world, rdzv)
    try:
        res = _drive(comm, rank, world)
        # host collectives of the product channel
        assert comm.all_reduce_max(rank * 10) == (world - 1) * 10
        assert comm.enable_rccl() is False and comm.rccl_error  # no GPU here: every rank stays on the host channel
        # failure agreement: rank 1 fails, both ranks raise together instead of rank 0 waiting in the next collective
        try:
            comm.raise_if_any(RuntimeError("boom") if rank == 1 else None, "step failed")
            res["agreed"] = False
        except Exception as e:
            res["agreed"] = "boom" in str(e)
        comm.barrier()
        if rank == 0:
            q.put(res)
    finally:
        comm.close()


def _worker_gloo(rank, world, port, q):
    import torch.distributed as td

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _drive(GlooComm(), rank, world)
        if rank == 0:
            q.put(res)
    finally:
        td.barrier()
        td.destroy_process_group()


def _check(res):
    N, E, obs_ens, obs, perturbs, decorr, taper = _inputs()
    assert np.abs(res["global"] - es.ens_update0(E, obs_ens, obs, perturbs, decorr)).max() < 1e-11
    assert np.abs(res["local"] - es.ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper)).max() < 1e-11
    perms = np.arange(N * 5, dtype=float).reshape(N, 5)
    assert np.array_equal(res["w"], (perms * 2 + perms * 0.5)[:, None, :])
    assert res["p"].shape == (N, 3, 4)


def _run(target, world, extra):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, extra, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_world_size_n_host_channel_sharded_update_and_forward(world):
    with tempfile.TemporaryDirectory() as d:
        res = _run(_worker_sockets, world, os.path.join(d, "rdzv"))
    _check(res)
    assert res["agreed"] is True


@pytest.mark.timeout(300)
def test_world_size_2_gloo_sharded_update_and_forward():
    _check(_run(_worker_gloo, 2, _free_port()))


def test_single_process_comm_is_identity():
    comm = Comm()
    a = np.arange(6.0)
    assert comm.world_size == 1 and comm.all_reduce_sum(a) is a and comm.all_gather_rows(a) is a
    N, E, obs_ens, obs, perturbs, decorr, _ = _inputs()
    out = sharded_update(HostShardPlan(N, E, obs_ens, obs, perturbs, decorr), comm)
    assert np.abs(out - es.ens_update0(E, obs_ens, obs, perturbs, decorr)).max() < 1e-11
    with pytest.raises(ValueError):
        forward_model_sharded(lambda a, b: [a, b], np.zeros((4, 2)), np.zeros((3, 2)), comm=comm)


def _worker_rdzv_hardening(rank, world, rdzv, q):
    """Rank 0 publishes; rank 1 inspects the file before connecting (mode, no key when the launcher passed it in the
    environment).  Then the RCCL pre-phase: rank 0 pretends it could initialise (a fake context whose hm_comm_create would
    block for ever), rank 1 cannot -- both must come back with the refusal, nobody enters the create call."""
    import stat
    import time as _t

    from historymatching_amd import _lib

    if rank == 1:
        t_end = _t.time() + 60
        while not os.path.exists(rdzv) and _t.time() < t_end:
            _t.sleep(0.01)
        st = os.stat(rdzv)
        mode_ok = stat.S_IMODE(st.st_mode) == 0o600 and st.st_uid == os.geteuid()
        key_in_file = open(rdzv).read().split()[1]
    comm = Comm(rank, world, rdzv)
    try:
        if rank == 0:
            class FakeLib:
                def hm_comm_unique_id(self, buf):
                    return 0

                def hm_comm_create(self, *a):
                    _t.sleep(3600)

                def hm_last_error(self):
                    return b""

            class FakeCtx:
                lib, handle, device = FakeLib(), None, 0

            ok = comm.enable_rccl(ctx=FakeCtx())
        else:
            ok = comm.enable_rccl()
        res = comm.host.gather((ok, comm.rccl_error, None if rank == 0 else (mode_ok, key_in_file)))
        if rank == 0:
            q.put(res)
    finally:
        comm.close()


@pytest.mark.timeout(120)
def test_rendezvous_file_is_private_and_rccl_refusal_is_agreed_before_init(monkeypatch):
    monkeypatch.setenv("HM_AMD_RDZV_KEY", "00112233445566778899aabbccddeeff")
    with tempfile.TemporaryDirectory() as d:
        res = _run(_worker_rdzv_hardening, 2, os.path.join(d, "rdzv"))
    (ok0, err0, _), (ok1, err1, (mode_ok, key_in_file)) = res
    assert ok0 is False and ok1 is False and "rank 1" in err0 and err0 == err1
    assert mode_ok and key_in_file == "-"


def test_peer_refuses_a_rendezvous_file_that_is_not_private(tmp_path):
    from historymatching_amd.dist import HostChannel

    f = tmp_path / "rdzv"
    f.write_text("1 00\n")
    os.chmod(f, 0o644)
    with pytest.raises(TimeoutError, match="not a private file"):
        HostChannel(1, 2, str(f), timeout=0.3)


def test_default_rendezvous_directory_is_private():
    from historymatching_amd.dist import default_rdzv_dir

    st = os.stat(default_rdzv_dir())
    assert st.st_uid == os.geteuid() and not (st.st_mode & 0o077)


@pytest.mark.timeout(120)
def test_bench_gpus_2_launches_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` run plainly, as the driver runs `--gpus 1`: the script becomes the launcher (two child
    processes, before any GPU call), the ranks rendezvous, one JSON line with n_gpus = 2 comes out.  --dry-run: no device."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HM_AMD_RDZV", "HM_AMD_RDZV_KEY")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=env, timeout=100)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] and line["ranks"]["self_launched"]
    assert sorted(s[0] for s in line["ranks"]["seen"]) == [0, 1] and len({s[2] for s in line["ranks"]["seen"]}) == 2
    # a launcher that set a different WORLD_SIZE: no line for the wrong rank count
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=100)
    assert r.returncode == 4
