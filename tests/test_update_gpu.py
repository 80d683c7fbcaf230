"""GPU parity tests of the ensemble-smoother update against (a) fixtures captured from the REAL reference
functions (tests/golden, oracle/make_golden.py) and (b) the oracle restatement on seeded inputs.

Tolerances: fp64 <= 1e-10 abs (min-flop association + sweep/Cholesky inverse instead of the
reference's left-to-right products + pinv: observed ~1e-13); fp32 <= 1e-4 relative to max |increment|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hm(golden):
    f1, f2, f3, f4 = (np.load(golden / n) for n in ("f1_rng_replay.npz", "f2_obs_error.npz", "f3_ens_update0.npz", "f4_ens_update0_loc.npz"))
    kw = dict(obs_ens=f3["obs_ens"], obs=f3["obs"], perturbs=f1["hm_perturbs"], decorr=f2["decorr"])
    return f1, f3, f4, kw


def test_center_matches_reference_fixture(golden):
    from historymatching_amd.update import center

    f5 = np.load(golden / "f5_helpers.npz")
    X, x = center(f5["a"])
    assert np.abs(X - f5["center_X"]).max() < 1e-15 and np.abs(x - f5["center_x"]).max() < 1e-15
    Xr, _ = center(f5["a"], rescale=True)
    assert np.abs(Xr - f5["center_Xr"]).max() < 1e-15


def test_ens_update0_gaussian_gaussian_fixture(golden):
    """The reference's bug check (HistoryMatch.py:594-612) through the GPU path."""
    from historymatching_amd.update import ens_update0

    f1, f3, _, _ = _hm(golden)
    E = f1["gg_E"]
    E0 = E.copy()
    post = ens_update0(E, E, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3))
    assert np.array_equal(E, E0)  # inputs unmodified
    assert np.abs(post - f3["gg_postr"]).max() < 1e-10
    assert np.allclose(post.mean(0), [0.98474734, 1.08434752, 1.02224568], atol=1e-8)


def test_ens_update0_history_matching_fixture(golden):
    from historymatching_amd.update import ens_update0

    f1, f3, _, kw = _hm(golden)
    assert np.abs(ens_update0(f1["perm_prior"], **kw) - f3["perm_es"]).max() < 1e-10
    assert np.abs(ens_update0(f3["obs_ens"], **kw) - f3["es0"]).max() < 1e-10  # M = n_obs (HistoryMatch.py:1156)


def test_ens_update0_fp32_tolerance(golden):
    from historymatching_amd.update import ens_update0

    f1, f3, _, kw = _hm(golden)
    out = ens_update0(f1["perm_prior"], **kw, dtype=32)
    assert out.dtype == np.float32
    inc = np.abs(f3["perm_es"] - f1["perm_prior"]).max()
    # NumPy run on fp32 copies of the same inputs is 8e-6 * inc off here (cond(C) = 1.7e4)
    assert np.abs(out - f3["perm_es"]).max() <= 1e-4 * inc


def test_ens_update0_loc_fixtures(golden):
    from historymatching_amd.update import ens_update0_loc

    f1, f3, f4, kw = _hm(golden)
    E = f1["gg_E"]
    gg = ens_update0_loc(E, E, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), np.eye(3))
    assert np.abs(gg - f4["gg_postr_loc"]).max() < 1e-10  # HistoryMatch.py:811-815
    ones = ens_update0_loc(f1["perm_prior"], **kw, taper=np.ones((400, 160)))
    assert np.abs(ones - f4["les_ones"]).max() < 1e-10
    assert np.allclose(ones, f3["perm_es"])  # "Reproduces global analysis?" HistoryMatch.py:821-822
    les = ens_update0_loc(f1["perm_prior"], **kw, taper=f4["taper"])
    assert np.abs(les - f4["perm_les"]).max() < 1e-10  # HistoryMatch.py:863


def test_loc_elements_with_no_obs_in_range_unchanged(golden):
    from historymatching_amd.update import ens_update0_loc

    f1, _, f4, kw = _hm(golden)
    taper = f4["taper"].copy()
    taper[:50] = 0.0
    taper[50:60] = 5e-5  # sqrt = 7e-3 < 1e-2 cutoff (HistoryMatch.py:786)
    out = ens_update0_loc(f1["perm_prior"], **kw, taper=taper)
    assert np.array_equal(out[:, :60], f1["perm_prior"][:, :60])
    assert np.abs(out[:, 60:] - f1["perm_prior"][:, 60:]).max() > 1e-3


def _seeded(N, M, n_obs, seed):
    from oracle import es

    rng = np.random.RandomState(seed)
    E = rng.randn(N, M) * 0.7 + rng.randn(M) * 3
    H = rng.randn(M, n_obs) / np.sqrt(M)
    obs_ens = E @ H
    _, R12, decorr = es.obs_error_model(40, 4) if n_obs == 160 else es.obs_error_model(n_obs, 1)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    return E, obs_ens, obs, perturbs, decorr


@pytest.mark.parametrize("N,M,n_obs", [(100, 400, 160), (37, 1000, 23), (1000, 4099, 160)])
def test_ens_update0_seeded_vs_oracle(N, M, n_obs):
    from historymatching_amd.update import ens_update0
    from oracle import es

    args = _seeded(N, M, n_obs, N + M)
    ref = es.ens_update0(*args)
    out = ens_update0(*args)
    assert np.abs(out - ref).max() < 1e-10
    out32 = ens_update0(*args, dtype=32)
    assert np.abs(out32 - ref).max() <= 1e-4 * np.abs(ref - args[0]).max()


def test_ens_update0_loc_seeded_vs_oracle():
    from historymatching_amd.update import ens_update0_loc
    from oracle import es

    rng = np.random.RandomState(9)
    N, M, n_obs = 60, 96, 48
    E = rng.randn(N, M)
    obs_ens = E @ (rng.randn(M, n_obs) / 10)
    _, R12, decorr = es.obs_error_model(12, 4)
    obs = obs_ens[3] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.3)
    ref = es.ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper)
    out = ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper)
    assert np.abs(out - ref).max() < 1e-10
    out32 = ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper, dtype=32)
    assert np.abs(out32 - ref).max() <= 1e-4 * np.abs(ref - E).max()


def test_update_properties_at_c3_size():
    """BASELINE config 3 shape (N=1000, M=128*128, n_obs=160), size-independent properties:
    (1) a zero-innovation update (D = 0) is the identity; (2) the update is affine-equivariant in the state:
    update(a*E + b) == a*update(E) + b for a column-wise shift b (anomalies remove it); (3) row-sharded phases
    with summed reduce buffers reproduce the single-shot update (the multi-GPU path of SURVEY.md 8e)."""
    from historymatching_amd.dist import sharded_update
    from historymatching_amd.update import UpdatePlan, ens_update0

    rng = np.random.RandomState(1)
    N, M, n_obs = 1000, 128 * 128, 160
    E = rng.randn(N, M)
    obs_ens = rng.rand(N, n_obs)
    decorr = np.eye(n_obs) * 3.0
    obs = rng.rand(n_obs)
    out = ens_update0(E, obs_ens, obs, obs - obs_ens, decorr)  # obs - obs_ens - perturbs == 0
    assert np.abs(out - E).max() < 1e-12
    perturbs = rng.randn(N, n_obs) * 0.1
    full = ens_update0(E, obs_ens, obs, perturbs, decorr)
    shift = rng.randn(M)
    full2 = ens_update0(2.0 * E + shift, obs_ens, obs, perturbs, decorr)
    assert np.abs(full2 - (2.0 * full + shift)).max() < 1e-9
    # (3) two uneven shards; the "all-reduce" is a host sum over the two plans
    plans = []
    for sl in (slice(0, 600), slice(600, N)):
        p = UpdatePlan(N, sl.stop - sl.start, M, n_obs)
        p.set_inputs(E[sl], obs_ens[sl], obs, perturbs[sl], decorr)
        plans.append(p)
    for ph in range(3):
        for p in plans:
            p.phase(ph)
        if ph < 2:
            for which in UpdatePlan.REDUCE_AFTER_PHASE[ph]:
                tot = sum(p.get_reduce(which) for p in plans)
                for p in plans:
                    p.set_reduce(which, tot)
    outs = np.concatenate([(p.sync(), p.output())[1] for p in plans])
    assert np.abs(outs - full).max() < 1e-11
    # single-rank driver path
    p = UpdatePlan(N, N, M, n_obs)
    p.set_inputs(E, obs_ens, obs, perturbs, decorr)
    assert np.abs(sharded_update(p) - full).max() < 1e-12


def test_es_mda_driver_matches_oracle_loop():
    """BASELINE config 3 in miniature: ES-MDA (n_iter passes of forward run + ens_update0 with alpha = n_iter,
    SURVEY.md 8f) through the GPU forward model and the GPU update vs the same loop through the oracle."""
    from historymatching_amd.forward import make_forward_model
    from historymatching_amd.update import es_mda
    from oracle import es
    from oracle.ressim import forward_model as oracle_forward
    from tests.helpers import make_models, perms

    nTime, N = 40, 16  # producers see water only late in the 40 steps (HistoryMatch.py:533-535)
    om, gm = make_models(20, 20)
    x = perms(20, 20, N + 1, seed=21, scale=0.5)
    truth, prior = x[0], x[1:]
    _, R12, _ = es.obs_error_model(nTime, 4)
    fm = make_forward_model(gm, 0.025, nTime, return_history=False)
    fwd_gpu = lambda E: es.vect(fm(E)[1], nTime)  # noqa: E731
    fwd_cpu = lambda E: es.vect(oracle_forward(om, E, None, 0.025, nTime)[1], nTime)  # noqa: E731
    obs = fwd_cpu(truth[None])[0] + R12 @ np.random.RandomState(5).randn(4 * nTime)
    post_gpu = es_mda(fwd_gpu, prior, obs, R12, n_iter=2, rng=np.random.RandomState(7))
    post_cpu = es.es_mda(fwd_cpu, prior, obs, R12, n_iter=2, rng=np.random.RandomState(7))
    assert post_gpu.shape == prior.shape
    assert np.abs(post_gpu - post_cpu).max() < 1e-6
    assert np.abs(post_gpu - prior).max() > 1e-3  # it did update


@pytest.mark.parametrize("dtype,tol", [(64, 1e-9), (32, 2e-3)])
def test_es_mda_device_resident_matches_host_driver(dtype, tol):
    """The device-resident ES-MDA (forward -> update -> forward chained in HBM: hm_fwd_set_inputs_device,
    hm_upd_set_inputs_device, hm_upd_swap) reproduces the host-driven loop pass for pass (same RNG stream)."""
    from historymatching_amd.forward import make_forward_model
    from historymatching_amd.update import es_mda, es_mda_device
    from oracle import es
    from tests.helpers import make_models, perms

    nTime, N = 40, 16
    _, gm = make_models(20, 20)
    x = perms(20, 20, N + 1, seed=21, scale=0.5)
    truth, prior = x[0], x[1:]
    _, R12, _ = es.obs_error_model(nTime, 4)
    fm = make_forward_model(gm, 0.025, nTime, return_history=False)
    fwd_gpu = lambda E: es.vect(fm(E)[1], nTime)  # noqa: E731
    obs = fwd_gpu(truth[None])[0] + R12 @ np.random.RandomState(5).randn(4 * nTime)
    host = es_mda(fwd_gpu, prior, obs, R12, n_iter=2, rng=np.random.RandomState(7), dtype=64)
    st = {}
    dev = es_mda_device(gm, prior, obs, R12, 0.025, nTime, n_iter=2, rng=np.random.RandomState(7), dtype=dtype, stats=st)
    assert dev.shape == prior.shape and st["ms_forward"] > 0 and st["ms_update"] > 0
    assert np.abs(dev - host).max() < tol * max(1.0, np.abs(host - prior).max())
    assert np.abs(dev - prior).max() > 1e-3  # it did update


@pytest.mark.parametrize("N,M,n_obs,localized", [(1000, 4096, 160, False), (999, 4100, 160, False), (130, 1024, 64, True),
                                                  (64, 512, 48, False), (200, 2048, 32, False), (300, 1024, 96, False),
                                                  (257, 1536, 128, False), (5, 256, 32, False), (17, 320, 160, False)])
def test_fp32_matrix_core_path_matches_generic_and_oracle(N, M, n_obs, localized):
    """fp32 update: v_mfma_f32_32x32x2 kernels (default) vs the generic VALU GEMMs (use_mfma=0) vs the fp64 oracle.
    Shapes cover exact tiles, ragged N / M, an n_obs that is not a multiple of 32 (falls back), and 2, 6, 8 and 10 block columns
    in the factorisation + gain launch of the fused run (spdinv.hip: k_ldl_chain)."""
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    rng = np.random.RandomState(N + M)
    E = rng.randn(N, M) + rng.randn(M)
    obs_ens = E[:, :n_obs] * 0.3 + rng.randn(N, n_obs) * 0.1
    _, R12, decorr = es.obs_error_model(n_obs // 4, 4)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.3) if localized else None
    ref = (es.ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper) if localized
           else es.ens_update0(E, obs_ens, obs, perturbs, decorr))
    outs = []
    for use_mfma in (1, 0):
        p = UpdatePlan(N, N, M, n_obs, dtype=32, localized=localized)
        p.set_option("use_mfma", use_mfma)
        p.set_inputs(E, obs_ens, obs, perturbs, decorr, taper)
        p.run_local()
        outs.append(p.output())
        p.close()
    inc = np.abs(ref - E).max()
    assert np.abs(outs[0] - ref).max() <= 1e-4 * inc
    assert np.abs(outs[1] - ref).max() <= 1e-4 * inc
    assert np.abs(outs[0].astype(np.float64) - outs[1]).max() <= 2e-5 * inc


def test_fused_run_is_bit_reproducible():
    """The chain in front of the apply hands tiles from wave to wave and from workgroup to workgroup through flags (spdinv.hip:
    sweeper, pivot wave, tile waves; the gain's workgroups taking block columns as they are published): every sum has a fixed
    order, so repeated runs of one plan give the same bits (300 repeats at three shapes: profiles/diag/upd_repeat.py)."""
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    N, M, n_obs = 600, 4096, 160
    rng = np.random.RandomState(11)
    _, R12, decorr = es.obs_error_model(n_obs // 4, 4)
    p = UpdatePlan(N, N, M, n_obs, dtype=32)
    p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, decorr)
    p.run_local()
    ref = p.output().copy()
    for _ in range(25):
        p.run_local()
        assert np.array_equal(p.output(), ref)
    p.close()


def test_gain_chain_variants_agree():
    """The fused run's chain in front of the apply in its forms -- explicit inverse + product (ldl_gain 0), factorisation and gain as
    two kernels (2), both in one launch with the gain's forward sweep running behind the factorisation (1, default) -- on the same
    inputs: the same update to fp32 rounding of the gain."""
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    N, M, n_obs = 500, 2048, 160
    rng = np.random.RandomState(7)
    E = rng.randn(N, M) + rng.randn(M)
    obs_ens = E[:, :n_obs] * 0.3 + rng.randn(N, n_obs) * 0.1
    _, R12, decorr = es.obs_error_model(n_obs // 4, 4)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    ref = es.ens_update0(E, obs_ens, obs, perturbs, decorr)
    inc = np.abs(ref - E).max()
    outs = {}
    for variant in (0, 2, 1):
        p = UpdatePlan(N, N, M, n_obs, dtype=32)
        p.set_option("ldl_gain", variant)
        p.set_inputs(E, obs_ens, obs, perturbs, decorr)
        for _ in range(3):  # the one-launch form keeps device counters across runs of a plan
            p.run_local()
        outs[variant] = p.output().astype(np.float64)
        p.close()
        assert np.abs(outs[variant] - ref).max() <= 1e-4 * inc
    assert np.abs(outs[1] - outs[0]).max() <= 2e-6 * inc
    assert np.array_equal(outs[1], outs[2])  # same factors, same tile products: the launch structure does not change a bit


@pytest.mark.parametrize("n", [16, 48, 160, 256])
def test_matrix_core_spd_inverse(n, golden):
    """spdinv.hip (rank-16 panels on the fp64 matrix cores, the C^-1 of fp32 plans) against numpy on C = S^T S + (N-1) I:
    random S, and at n = 160 the reference's own case (cond 1.7e4)."""
    import ctypes as C

    from historymatching_amd import _lib

    ctx, lib = _lib.Context.get(0), _lib.load()
    rng = np.random.RandomState(n)
    cases = [(rng.randn(40, n), 39.0)]
    if n == 160:
        _, f3, _, kw = _hm(golden)
        Y = kw["obs_ens"] - kw["obs_ens"].mean(0)
        cases.append((Y @ kw["decorr"], float(Y.shape[0] - 1)))
    for S, ridge in cases:
        G = np.ascontiguousarray(S.T @ S)
        W = np.empty_like(G)
        dp = C.POINTER(C.c_double)
        _lib.check(lib.hm_debug_spd_inverse(ctx.handle, n, G.ctypes.data_as(dp), ridge, W.ctypes.data_as(dp)), "hm_debug_spd_inverse")
        ref = np.linalg.inv(G + ridge * np.eye(n))
        assert np.abs(W - ref).max() <= 1e-9 * np.abs(ref).max()
        assert np.abs(W - W.T).max() <= 1e-9 * np.abs(ref).max()  # off-diagonal tiles are mirrored, diagonal tiles swept


@pytest.mark.parametrize("n,N", [(16, 5), (48, 64), (160, 1000), (176, 130)])
def test_ldl_gain_matches_numpy(golden, n, N):
    """The gain of the fused analysis step, D0 B^-1, comes from a block L D L^T factorisation of B and tile products with its
    factors (spdinv.hip: k_ldl_factor, k_ldl_gain) instead of an explicit inverse: against NumPy in fp64, on a random SPD matrix and
    (n = 160) on the reference's own C = S^T S + (N-1) I; the result is stored in fp32."""
    import ctypes as C

    from historymatching_amd import _lib

    ctx, lib = _lib.Context.get(0), _lib.load()
    rng = np.random.RandomState(n + N)
    cases = [(rng.randn(40, n), 39.0)]
    if n == 160:
        _, f3, _, kw = _hm(golden)
        Y = kw["obs_ens"] - kw["obs_ens"].mean(0)
        cases.append((Y @ kw["decorr"], float(Y.shape[0] - 1)))
    for S, ridge in cases:
        G = np.ascontiguousarray(S.T @ S)
        X = np.ascontiguousarray(rng.randn(N, n))
        A_T = np.empty((n, N), dtype=np.float32)
        dp = C.POINTER(C.c_double)
        _lib.check(lib.hm_debug_ldl_gain(ctx.handle, n, N, G.ctypes.data_as(dp), ridge, X.ctypes.data_as(dp),
                                         A_T.ctypes.data_as(C.POINTER(C.c_float))), "hm_debug_ldl_gain")
        ref = X @ np.linalg.inv(G + ridge * np.eye(n))
        assert np.abs(A_T.T - ref).max() <= 2e-7 * np.abs(ref).max()  # fp32 storage of an fp64 result


def test_rccl_communicator_runs_the_analysis_step_on_library_buffers():
    """The multi-rank update's collectives are issued by the library itself (hm_comm_* / hm_upd_run_comm: RCCL opened with
    dlopen, in place on the plan's device buffers, on the context's stream).  One rank here (RCCL refuses two ranks on one
    GPU and the test box has one): a real RCCL communicator of world size 1 drives (a) MAX / SUM all-reduces and an
    all-gather on library memory, (b) the whole global analysis step (fp32 matrix-core path and fp64) and (c) the localised
    one, each equal to the same plan's single-process run.  The two-rank arithmetic is covered on the host channel
    (tests/test_dist_cpu.py, tests/test_dist_gpu.py)."""
    import ctypes as C

    from historymatching_amd import _lib
    from historymatching_amd.dist import Comm, sharded_update
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    comm = Comm()
    assert comm.enable_rccl() is False                      # one rank: nothing to create ...
    assert comm.enable_rccl(force_single=True), comm.rccl_error   # ... unless asked to
    lib, ctx = comm.ctx.lib, comm.ctx
    assert lib.hm_comm_world_size(comm.rccl) == 1 and lib.hm_comm_rank(comm.rccl) == 0
    rng = np.random.RandomState(3)
    N, M, n_obs = 96, 512, 32
    E = rng.randn(N, M) + rng.randn(M)
    obs_ens = 0.3 * E[:, :n_obs] + 0.1 * rng.randn(N, n_obs)
    _, R12, decorr = es.obs_error_model(n_obs // 4, 4)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.3)
    # (a) raw collectives on a plan's reduce buffer
    p = UpdatePlan(N, N, M, n_obs, dtype=32)
    p.set_inputs(E, obs_ens, obs, perturbs, decorr)
    p.phase(0)
    before = p.get_reduce(0).copy()
    ptr, n, dt = p.reduce_buffer(0)
    comm.device_all_reduce(ptr, n, dt, "sum")
    comm.device_all_reduce(ptr, n, dt, "max")
    _lib.check(lib.hm_comm_all_gather(comm.rccl, C.c_void_p(ptr), n, 32), "hm_comm_all_gather")
    _lib.check(lib.hm_comm_broadcast(comm.rccl, C.c_void_p(ptr), n, 32, 0), "hm_comm_broadcast")
    comm.device_sync()
    assert np.array_equal(p.get_reduce(0), before)
    p.close()
    # (b), (c): the analysis step through hm_upd_run_comm
    for dtype, loc in ((32, False), (64, False), (32, True), (64, True)):
        outs = []
        for through_comm in (True, False):
            p = UpdatePlan(N, N, M, n_obs, dtype=dtype, localized=loc)
            p.set_inputs(E, obs_ens, obs, perturbs, decorr, taper if loc else None)
            if through_comm:
                outs.append(sharded_update(p, comm))
                assert p.ctx is ctx
            else:
                for ph in range(3):
                    p.phase(ph)
                p.sync()
                outs.append(p.output())
            p.close()
        assert np.array_equal(outs[0], outs[1])
        ref = es.ens_update0_loc(E, obs_ens, obs, perturbs, decorr, taper) if loc else es.ens_update0(E, obs_ens, obs, perturbs, decorr)
        assert np.abs(outs[0] - ref).max() <= (1e-10 if dtype == 64 else 1e-4 * np.abs(ref - E).max())
    comm.close()


def test_column_sharded_localised_solves_match_unsharded():
    """the per-element solves of the localised analysis sharded by state column over G ranks (here G plans on one
    GPU, the all-gather of the weights done on the host) reproduce the unsharded plan bit for bit -- M not a multiple of G, so
    the last block is ragged and the buffer padded."""
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    rng = np.random.RandomState(12)
    N, M, n_obs, G = 64, 1001, 32, 3
    E = rng.randn(N, M)
    obs_ens = 0.3 * E[:, :n_obs] + 0.1 * rng.randn(N, n_obs)
    _, R12, decorr = es.obs_error_model(n_obs // 4, 4)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.3)
    for dtype in (64, 32):
        full = UpdatePlan(N, N, M, n_obs, dtype=dtype, localized=True)
        full.set_inputs(E, obs_ens, obs, perturbs, decorr, taper)
        ref = (full.run_local(), full.output())[1]
        full.close()
        plans = []
        for r in range(G):  # every "rank" holds all rows here: only the column shard differs
            p = UpdatePlan(N, N, M, n_obs, dtype=dtype, localized=True)
            p.set_inputs(E, obs_ens, obs, perturbs, decorr, taper)
            p.set_column_shard(r, G)
            for ph in range(3):
                p.phase(ph)
            plans.append(p)
        blocks = [p.get_reduce(4).reshape(G, -1)[r] for r, p in enumerate(plans)]
        for p in plans:
            p.set_reduce(4, np.concatenate(blocks))
            p.phase(3)
            p.sync()
            assert np.array_equal(p.output(), ref)
            p.close()


def test_ies_matches_reference_fixtures(golden):
    """the iterative ensemble smoother (subspace algebra on the host, centring / re-composition on
    the GPU) against outputs of the REAL reference `IES` (tests/golden/f6_iterative.npz): the linear-Gaussian bug check
    (must also reproduce the non-iterative analysis, HistoryMatch.py:949-951) and a 3-iterate run with a linear
    observation operator."""
    from historymatching_amd.update import ies, recompose

    f1, f3, _, kw = _hm(golden)
    f6 = np.load(golden / "f6_iterative.npz")
    post, stats = ies(f1["gg_E"], lambda x: x, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), subspace="gram")
    assert len(stats["E"]) == 4 and stats["Eo"][0].shape == (400, 3)
    assert np.abs(post - f6["ies_gg"]).max() < 1e-9
    assert np.allclose(post, f3["gg_postr"])
    H = f3["H"]
    post, _ = ies(f1["perm_prior"], lambda x: x @ H, kw["obs"], kw["perturbs"], kw["decorr"], xStep=0.4, iMax=3)
    assert np.abs(post - f6["ies_lin"]).max() < 1e-8
    # the GPU re-composition alone
    rng = np.random.RandomState(2)
    W, X0, x0 = rng.randn(37, 37), rng.randn(37, 301), rng.randn(301)
    assert np.abs(recompose(W, X0, x0) - (x0 + W @ X0)).max() < 1e-12
    assert np.abs(recompose(W, X0, x0, dtype=32) - (x0 + W @ X0)).max() < 1e-3


def test_ies_device_subspace_step_matches_host_and_fixtures(golden):
    """`ies(subspace="device")`: the Gauss-Newton step of the weights on the GPU (the localised smoother's device step with one
    domain and a taper of ones) against the REAL reference's outputs (F6) and against the host form on a nonlinear operator with
    partial steps."""
    from historymatching_amd.update import ies
    from oracle import es

    f1, f3, _, kw = _hm(golden)
    f6 = np.load(golden / "f6_iterative.npz")
    post, stats = ies(f1["gg_E"], lambda x: x, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), subspace="device")
    assert len(stats["E"]) == 4
    assert np.abs(post - f6["ies_gg"]).max() < 1e-9
    H = f3["H"]
    post, _ = ies(f1["perm_prior"], lambda x: x @ H, kw["obs"], kw["perturbs"], kw["decorr"], xStep=0.4, iMax=3, subspace="device")
    assert np.abs(post - f6["ies_lin"]).max() < 1e-8
    rng = np.random.RandomState(23)
    N, M, n_obs = 60, 90, 12
    E = rng.randn(N, M)
    Hn = rng.randn(M, n_obs) / 8
    fwd = lambda x: np.tanh(x @ Hn) + 0.1 * (x @ Hn) ** 2  # noqa: E731
    _, R12, decorr = es.obs_error_model(3, 4)
    obs = fwd(E[:1])[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    dev, sd = ies(E, fwd, obs, perturbs, decorr, xStep=0.6, iMax=3, subspace="device")
    host, sh = ies(E, fwd, obs, perturbs, decorr, xStep=0.6, iMax=3, subspace="gram")
    assert np.abs(dev - host).max() < 1e-9 and np.abs(dev - E).max() > 1e-2
    for a, b in zip(sd["Eo"], sh["Eo"]):
        assert np.abs(a - b).max() < 1e-9


@pytest.mark.parametrize("nx,ny,N", [(20, 20, 7), (128, 128, 5), (96, 160, 3)])
def test_device_kronecker_prior_sampler_matches_host(nx, ny, N):
    """hm_sample_kron: the separable prior sampler with its two contractions on the fp64 matrix cores
    reproduces the host NumPy sampler for the same seed to rounding, and can leave the fields in a forward plan's device
    buffer.  The law itself (Cov = Cx (x) Cy == the reference dense covariance) is pinned in test_oracle_golden."""
    from historymatching_amd.geostat import gaussian_fields_kron, gaussian_fields_kron_device

    host = gaussian_fields_kron(nx, ny, 2, 1, N, r=0.8, seed=11)
    dev = gaussian_fields_kron_device(nx, ny, 2, 1, N, r=0.8, seed=11)
    assert dev.shape == host.shape
    assert np.abs(dev - host).max() <= 1e-12 * max(1.0, np.abs(host).max())


def test_device_prior_sampler_has_the_reference_law(golden):
    """hm_sample_kron against the REFERENCE's prior law (fixture F11: rows of the dense covariance the reference's geostat.py forms on the
    default 20 x 20 grid, and the 2000-sample covariance of its own sampler): the sample covariance of 40 000 device-drawn fields
    matches the reference's covariance rows within sampling error (4 standard errors of a covariance estimate at this size: 0.03; the
    reference's own sampler is 0.11 off at its 2000 samples), their mean is zero and their variance one -- and an oracle-side dense
    sampler (oracle/geostat.py, the reference's algorithm) gives the same statistics."""
    from historymatching_amd.geostat import gaussian_fields_kron_device
    from oracle import geostat as og

    f = np.load(golden / "f11_prior_law.npz")
    Nx, Ny, Lx, Ly, r, cells = int(f["Nx"]), int(f["Ny"]), float(f["Lx"]), float(f["Ly"]), float(f["r"]), f["cells"]
    n = 40000
    dev = gaussian_fields_kron_device(Nx, Ny, Lx, Ly, n, r=r, seed=123)
    assert dev.shape == (n, Nx * Ny) and np.abs(dev.mean(0)).max() < 5 / np.sqrt(n)
    X = dev - dev.mean(0)
    cov_rows = (X[:, cells].T @ X) / (n - 1)
    assert np.abs(cov_rows - f["cov_rows"]).max() < 0.03
    assert np.abs(np.abs(f["ref_sample_cov_2000"] - f["cov_rows"]).max() - 0.1) < 0.1   # (what 2000 samples of the reference's own sampler give)
    dense = og.gaussian_fields(Nx, Ny, Lx, Ly, 4000, r=r, rng=np.random.RandomState(5))
    Xd = dense - dense.mean(0)
    assert np.abs((Xd[:, cells].T @ Xd) / (len(Xd) - 1) - f["cov_rows"]).max() < 0.1


def test_device_iles_matches_reference_fixture_and_host_twin(golden):
    """the localised iterative smoother with its per-domain subspace algebra on the device (iles.hip: LU
    solve with W, n_loc x n_loc Cholesky, push-through form of the Gauss-Newton step) against
    (a) the output of the REAL reference `ILES` on the linear-Gaussian bug check (fixture F6 `iles_gg`: N = 400, one element per
        batch, taper = I), which also reproduces the non-iterative local analysis (HistoryMatch.py:1069-1071);
    (b) the host twin (pseudo-inverse + SVD per element, the reference's own evaluation order) on a nonlinear observation
        operator with partial steps, overlapping local domains and elements without any observation in range."""
    from historymatching_amd.update import iles, iles_host
    from oracle import es

    f1, f4, f6 = (np.load(golden / n) for n in ("f1_rng_replay.npz", "f4_ens_update0_loc.npz", "f6_iterative.npz"))
    post, stats = iles(f1["gg_E"], lambda x: x, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), taper=np.eye(3))
    assert len(stats["E"]) == 4
    assert np.abs(post - f6["iles_gg"]).max() < 1e-9
    assert np.allclose(post, f4["gg_postr_loc"])
    rng = np.random.RandomState(17)
    N, M, n_obs = 30, 40, 12
    E = rng.randn(N, M)
    H = rng.randn(M, n_obs) / 6
    fwd = lambda x: np.tanh(x @ H) + 0.1 * (x @ H) ** 2  # noqa: E731
    _, R12, decorr = es.obs_error_model(3, 4)
    obs = fwd(E[:1])[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    taper = es.bump(rng.rand(M, n_obs) * 1.5)
    taper[5] = 0.0   # an element with no observation in range keeps its prior weights
    dev, sd = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.6, iMax=3)
    host, sh = iles_host(E, fwd, obs, perturbs, decorr, taper, xStep=0.6, iMax=3)
    assert np.abs(dev - host).max() < 1e-9 and np.abs(dev[:, 5] - E[:, 5]).max() < 1e-14  # x0 + I X0: re-composed, not copied
    for a, b in zip(sd["Eo"], sh["Eo"]):
        assert np.abs(a - b).max() < 1e-9
    assert np.abs(dev - E).max() > 1e-2
    # the blocked form of the step (the default from N = 256 on: the fixture run above took it) on the small case: same elimination
    blk, _ = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.6, iMax=3, blocked=True)
    assert np.abs(blk - dev).max() < 1e-12
    one, _ = iles(f1["gg_E"], lambda x: x, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), taper=np.eye(3), blocked=False)
    assert np.abs(one - post).max() < 1e-12


def test_device_iles_blocked_form_equals_one_workgroup_form():
    """iles.hip has two forms of the Gauss-Newton step: one workgroup per domain (small ensembles) and, from N = 256 members on, a
    blocked elimination with partial pivoting over many workgroups per domain (panels of 16 columns, the products on the fp64
    matrix cores).  Same pivots, same sequence of multiply-adds per element in the elimination: the two agree to rounding of the
    products -- on an ensemble size that is not a multiple of the panel width, domains with different numbers of observations in
    range (one with none), partial steps, three iterates; and the re-composition on the matrix cores (domains of >= 16 elements)
    equals the scalar one."""
    from historymatching_amd.update import IlesPlan, ies, iles
    from oracle import es

    rng = np.random.RandomState(23)
    N, M, n_obs = 301, 90, 21
    E = rng.randn(N, M)
    H = rng.randn(M, n_obs) / 9
    fwd = lambda x: np.tanh(x @ H) + 0.1 * (x @ H) ** 2  # noqa: E731
    decorr = np.diag(1.0 + rng.rand(n_obs))
    obs = fwd(E[:1])[0] + 0.3 * rng.randn(n_obs)
    perturbs = 0.3 * rng.randn(N, n_obs)
    batches = [np.arange(0, 40), np.arange(40, 45), np.arange(45, 70), np.arange(70, 90)]
    taper = np.zeros((M, n_obs))
    for i, b in enumerate(batches):
        taper[b] = es.bump(rng.rand(n_obs) * (1.1 + 0.3 * i))
    taper[batches[1]] = 0.0  # a domain without any observation in range
    a, sa = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.6, iMax=3, batches=batches, blocked=True)
    b, sb = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.6, iMax=3, batches=batches, blocked=False)
    assert np.abs(a - b).max() < 1e-11 * max(1.0, np.abs(b).max()), np.abs(a - b).max()
    assert np.abs(a[:, 40:45] - E[:, 40:45]).max() < 1e-13 and np.abs(a - E).max() > 1e-2
    # one step from the identity, weights compared directly
    plans = []
    for blocked in (1, 0):
        plan = IlesPlan(E, batches, np.stack([taper[bb].mean(0) for bb in batches]))
        plan.set_option("blocked", blocked)
        Eo = fwd(E) @ decorr
        plan.step(Eo - Eo.mean(0), (obs - fwd(E) - perturbs) @ decorr, 0.8)
        plans.append([plan.weights(i) for i in range(len(batches))] + [plan.compose()])
        plan.close()
    for x, y in zip(*plans):
        assert np.abs(x - y).max() < 1e-12 * max(1.0, np.abs(y).max())
    assert np.array_equal(plans[0][1], np.eye(N))
    # IES through the one-domain device step at this ensemble size = the host algebra
    dev, _ = ies(E, fwd, obs, perturbs, decorr, xStep=0.5, iMax=3, subspace="device")
    host, _ = ies(E, fwd, obs, perturbs, decorr, xStep=0.5, iMax=3, subspace="gram")
    assert np.abs(dev - host).max() < 1e-10 * max(1.0, np.abs(host).max())
    auto, _ = ies(E, fwd, obs, perturbs, decorr, xStep=0.5, iMax=3)  # the default takes the device step from N = 256 on
    assert np.array_equal(auto, dev)


@pytest.mark.parametrize("blocked", [0, 1])
def test_device_iles_singular_weight_matrix_is_reported_and_left_alone(blocked):
    """A local domain whose weight matrix has become singular (here: written so by hand through the device pointer) cannot take the
    Gauss-Newton step (center(W^-1 .) does not exist): the step reports it (error 4, "singular") and leaves that domain's weights as
    they are, the other domains take their step -- in both forms of the step (one workgroup per domain / blocked elimination)."""
    import ctypes as C

    from historymatching_amd import _lib
    from historymatching_amd.update import IlesPlan

    rng = np.random.RandomState(31)
    N, M, n_obs = 40, 12, 7
    E = rng.randn(N, M)
    batches = [np.arange(0, 6), np.arange(6, 12)]
    plan = IlesPlan(E, batches, np.ones((2, n_obs)), cutoff=0.5)
    plan.set_option("blocked", blocked)
    Wbad = np.eye(N)
    Wbad[5] = 0.0
    ptr = plan.lib.hm_iles_device_ptr(plan.h, b"W")
    _lib.check(plan.lib.hm_copy_to_device(plan.ctx.handle, C.c_void_p(ptr + N * N * 8), Wbad.ctypes.data_as(C.c_void_p), Wbad.nbytes), "copy W")
    S = rng.randn(N, n_obs)
    S -= S.mean(0)
    D = rng.randn(N, n_obs)
    with pytest.raises(_lib.HmError, match="singular"):
        plan.step(S, D, 0.5)
    assert np.array_equal(plan.weights(1), Wbad)
    W0 = plan.weights(0)
    assert np.isfinite(W0).all() and np.abs(W0 - np.eye(N)).max() > 1e-3
    plan.close()


def test_device_iles_partitioned_domains():
    """The batched form (HistoryMatch.py:802-804, localization.py:95-145): elements of a rectangular domain share one weight
    matrix and the mean of their taper rows.  (a) With a taper that is constant inside every domain the batched run equals the
    per-element run (the M N^2 weight storage shrinks to B N^2 with no change of result); (b) a 128 x 128 state with N = 100
    members in 8 x 8-cell domains, simulated observations from the GPU forward model, three iterates: every iterate lowers the
    data mismatch, the posterior is finite and moved."""
    from historymatching_amd.forward import make_forward_model
    from historymatching_amd.localization import rectangular_partitioning, taper_for_wells
    from historymatching_amd.update import iles
    from oracle import es
    from tests.helpers import make_models, perms

    rng = np.random.RandomState(5)
    shape, N, n_obs = (12, 10), 24, 8
    M = shape[0] * shape[1]
    batches = rectangular_partitioning(shape, (4, 5))
    assert len(batches) == 6 and sum(len(b) for b in batches) == M
    taper = np.zeros((M, n_obs))
    for b in batches:
        taper[b] = es.bump(rng.rand(n_obs) * 1.4)
    E = rng.randn(N, M)
    H = rng.randn(M, n_obs) / 10
    fwd = lambda x: x @ H + 0.05 * (x @ H) ** 3  # noqa: E731
    decorr = np.eye(n_obs) * 2.0
    obs = fwd(E[:1])[0] + 0.5 * rng.randn(n_obs)
    perturbs = 0.5 * rng.randn(N, n_obs)
    per_elem, _ = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.7, iMax=3)
    batched, _ = iles(E, fwd, obs, perturbs, decorr, taper, xStep=0.7, iMax=3, batches=batches)
    assert np.abs(batched - per_elem).max() < 1e-10

    nTime, n, N = 40, 128, 100
    _, gm = make_models(n, n)
    x = perms(n, n, N + 1, seed=61, scale=0.6)
    truth, prior = x[0], x[1:]
    fm = make_forward_model(gm, 0.025, nTime, return_history=False)
    obs_fn = lambda E: es.vect(fm(E)[1], nTime)  # noqa: E731
    _, R12, decorr = es.obs_error_model(nTime, 4)
    obs = obs_fn(truth[None])[0] + R12 @ np.random.RandomState(3).randn(4 * nTime)
    perturbs = np.random.RandomState(4).randn(N, 4 * nTime) @ R12.T
    taper = taper_for_wells(gm, gm.xy2ind(*gm.prd_xy.T), nTime, radius=1.2)
    batches = rectangular_partitioning((n, n), (8, 8))
    assert len(batches) == 256
    post, stats = iles(prior, obs_fn, obs, perturbs, decorr, taper, xStep=0.4, iMax=3, batches=batches)
    assert post.shape == prior.shape and np.isfinite(post).all()
    mism = [np.sqrt(np.mean(((Eo - obs) @ decorr) ** 2)) for Eo in stats["Eo"]] + [np.sqrt(np.mean(((obs_fn(post) - obs) @ decorr) ** 2))]
    assert all(b < a for a, b in zip(mism, mism[1:])), mism
    assert np.abs(post - prior).max() > 1e-2
