"""Multi-rank ES-MDA with device-resident members (historymatching_amd.dist.es_mda_sharded) on the one GPU of the test box:
two processes (gloo rendezvous, host-staged all-reduces; both ranks use device 0) against the single-process
device-resident driver on the whole ensemble.  The RCCL form of the same all-reduces is covered by
test_update_gpu.py::test_rccl_all_reduce_on_library_buffers."""
import os
import socket

import numpy as np
import pytest

from tests.helpers import perms, wells_4corners

pytestmark = pytest.mark.gpu

DT, NT, NX, N = 0.025, 40, 20, 22


def _problem():
    from historymatching_amd.localization import taper_for_wells
    from historymatching_amd.obs import obs_error_model
    from historymatching_amd.ressim import ResSim

    model = wells_4corners(ResSim(NX, NX, 2, 1))
    prior = perms(NX, NX, N, seed=41)
    _, R12 = obs_error_model(NT, model.nPrd)
    obs = np.clip(0.3 + 0.05 * np.random.RandomState(8).randn(NT * model.nPrd), 0, 1)
    taper = taper_for_wells(model, model.xy2ind(*model.prd_xy.T), NT)
    return model, prior, obs, R12, taper


def _worker(rank, world, port, q):
    import torch.distributed as td

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from historymatching_amd.dist import Comm, es_mda_sharded, shard_bounds

        comm = Comm()
        model, prior, obs, R12, taper = _problem()
        lo, hi = shard_bounds(N, world, rank)
        res = {}
        for name, tp in (("global", None), ("local", taper)):
            post = es_mda_sharded(model, prior[lo:hi], obs, R12, DT, NT, n_iter=2, seed=5, comm=comm, dtype=64, taper=tp, device=0)
            res[name] = comm.all_gather_rows(post)
        if rank == 0:
            q.put(res)
    finally:
        td.barrier()
        td.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_rank_es_mda_matches_single_process():
    import torch.multiprocessing as mp

    from historymatching_amd.dist import es_mda_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = None
    for _ in range(500):  # a crashed worker must fail the test at once, not after the queue's timeout
        try:
            res = q.get(timeout=1.0)
            break
        except Exception:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(60 if res is not None else 5)
        if p.is_alive():
            p.terminate()
    assert res is not None and all(p.exitcode == 0 for p in procs)
    model, prior, obs, R12, taper = _problem()
    for name, tp in (("global", None), ("local", taper)):
        ref = es_mda_sharded(model, prior, obs, R12, DT, NT, n_iter=2, seed=5, dtype=64, taper=tp, device=0)
        assert res[name].shape == ref.shape
        # same members, same perturbations; only the order of the cross-rank sums differs
        assert np.abs(res[name] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        assert np.abs(ref - prior).max() > 1e-3  # the assimilation moved the ensemble


def test_single_rank_es_mda_sharded_equals_es_mda_device():
    from historymatching_amd.dist import es_mda_sharded
    from historymatching_amd.update import es_mda_device

    model, prior, obs, R12, _ = _problem()
    a = es_mda_sharded(model, prior, obs, R12, DT, NT, n_iter=2, seed=5, dtype=64, device=0)
    b = es_mda_device(model, prior, obs, R12, DT, NT, n_iter=2, rng=np.random.RandomState(5), dtype=64, device=0)
    assert np.array_equal(a, b)
