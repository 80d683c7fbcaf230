"""Multi-rank ES-MDA with device-resident members (historymatching_amd.dist.es_mda_sharded) on the one GPU of the test box:
two processes (the product's host channel; both ranks use device 0) against the single-process device-resident driver on
the whole ensemble.  Both ranks ask for RCCL first: RCCL does not accept two ranks of a communicator on one GPU
("Duplicate GPU detected"), which the ranks agree on and then run the same reductions host-staged.  The RCCL form itself
runs in test_update_gpu.py::test_rccl_communicator_runs_the_analysis_step_on_library_buffers."""
import os
import tempfile

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


def _worker(rank, world, rdzv, q):
    from historymatching_amd.dist import Comm, es_mda_sharded, shard_bounds

    comm = Comm(rank, world, rdzv, local_rank=0)
    try:
        got_rccl = comm.enable_rccl()
        model, prior, obs, R12, taper = _problem()
        lo, hi = shard_bounds(N, world, rank)
        res = {"rccl": got_rccl, "rccl_error": comm.rccl_error}
        for name, tp in (("global", None), ("local", taper)):
            post = es_mda_sharded(model, prior[lo:hi], obs, R12, DT, NT, n_iter=2, seed=5, comm=comm, dtype=64, taper=tp, device=0)
            res[name] = comm.all_gather_rows(post)
        comm.barrier()
        if rank == 0:
            q.put(res)
    finally:
        comm.close()


@pytest.mark.timeout(600)
def test_two_rank_es_mda_matches_single_process():
    import multiprocessing as mp

    from historymatching_amd.dist import es_mda_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_worker, args=(r, 2, os.path.join(d, "rdzv"), q)) for r in range(2)]
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
    # two ranks on ONE device: RCCL must have been refused on both ranks, with its own diagnosis
    if not res["rccl"]:
        assert "rank 0" in res["rccl_error"] and "rank 1" in res["rccl_error"]
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
