"""Shared builders for the parity tests (oracle side and GPU side use identical inputs)."""
import numpy as np

from historymatching_amd.geostat import gaussian_fields_kron


def wells_4corners(model):
    """Well layout of the reference (HistoryMatch.py:177-190)."""
    near01 = np.array([0.12, 0.87])
    model.prd_xy = [[x, y] for y in model.Ly * near01 for x in model.Lx * near01]
    model.inj_xy = [[model.Lx / 2, model.Ly / 2]]
    model.inj_rates = [[1]]
    model.prd_rates = np.ones((4, 1)) / 4
    return model


def make_models(Nx, Ny, dtype=64):
    """(oracle model, GPU-backed model) with identical grid and wells."""
    from historymatching_amd.ressim import ResSim as GpuResSim
    from oracle.ressim import ResSim as OracleResSim

    return wells_4corners(OracleResSim(Nx, Ny, 2, 1)), wells_4corners(GpuResSim(Nx, Ny, 2, 1, dtype=dtype))


def perms(Nx, Ny, N, seed=1, scale=1.0):
    return scale * gaussian_fields_kron(Nx, Ny, 2, 1, N, r=0.8, seed=seed)


def oracle_sim_and_noise(om, x, dt, nTime, wsat0=None):
    """Oracle simulation of one member plus the oracle's OWN numerical noise: the same restatement with
    SuperLU's column ordering switched from COLAMD to NATURAL.  The TPFA system is ill-conditioned
    (transmissibilities up to ~1e6: T*eps*|p| ~ 1e-9 flux noise for any fp64 solver), and the saturation
    front amplifies it, so parity of a different direct solver can only be asked to within this spread."""
    import oracle.ressim as orc
    from scipy.sparse.linalg import spsolve

    ref, noise, _ = oracle_sim_noise_and_nts(om, x, dt, nTime, wsat0)
    return ref, noise


def oracle_sim_noise_and_nts(om, x, dt, nTime, wsat0=None):
    """`oracle_sim_and_noise` plus the oracle's per-step CFL sub-step counts (Nts = ceil(dt / cfl) is a
    discontinuity; a whole-run comparison only means something where GPU and oracle took the same counts).  The counts of
    the two orderings are asserted equal: a member on which the oracle disagrees with itself about Nts cannot be a parity case."""
    import oracle.ressim as orc
    from scipy.sparse.linalg import spsolve

    orc.set_perm(om, x)
    w0 = np.zeros(om.Nxy) if wsat0 is None else wsat0
    ref = om.sim(dt, nTime, w0)
    nts = om.nts_trace.copy()
    orig = orc.spsolve
    orc.spsolve = lambda A, b: spsolve(A.tocsc(), b, permc_spec="NATURAL")
    try:
        ref2 = om.sim(dt, nTime, w0)
    finally:
        orc.spsolve = orig
    assert np.array_equal(nts, om.nts_trace), "the oracle's two solver orderings disagree on the sub-step counts"
    return ref, float(np.abs(ref2 - ref).max()), nts


def _oracle_sim_task(args):
    """One oracle simulation with a given SuperLU column ordering (worker of `oracle_sims_and_noise_parallel`)."""
    import oracle.ressim as orc
    from oracle.ressim import ResSim as OracleResSim
    from scipy.sparse.linalg import spsolve

    nx, ny, x, dt, nTime, permc = args
    om = wells_4corners(OracleResSim(nx, ny, 2, 1))
    orc.set_perm(om, x)
    if permc != "COLAMD":
        orc.spsolve = lambda A, b: spsolve(A.tocsc(), b, permc_spec=permc)
    return om.sim(dt, nTime, np.zeros(om.Nxy))


def oracle_sims_and_noise_parallel(nx, ny, xs, dt, nTime, permc2="NATURAL"):
    """`oracle_sim_and_noise` for several members at once, one process per simulation (the 512 x 512 oracle takes ~40 s
    per member-step, nearly all of it the 9 831 explicit sub-steps).  `permc2`: the second column ordering whose result
    measures the oracle's own solver noise (NATURAL fills in too much at 512 x 512: MMD_AT_PLUS_A there)."""
    import multiprocessing as mp

    tasks = [(nx, ny, x, dt, nTime, permc) for x in xs for permc in ("COLAMD", permc2)]
    with mp.get_context("spawn").Pool(min(len(tasks), 8)) as pool:
        res = pool.map(_oracle_sim_task, tasks, chunksize=1)
    return [(res[2 * m], float(np.abs(res[2 * m + 1] - res[2 * m]).max())) for m in range(len(xs))]


def _oracle_loc_task(args):
    import threadpoolctl

    from oracle import es

    with threadpoolctl.threadpool_limits(1):  # 8 BLAS threads are 25x slower on these tiny matrices
        return es.ens_update0_loc(*args)


def oracle_update_loc_columns(E_cols, obs_ens, obs, perturbs, decorr, taper_rows, nproc=8):
    """oracle.es.ens_update0_loc on a subset of state columns (each local analysis only reads its own column of E and
    its own row of the taper, HistoryMatch.py:783-793), the columns split over `nproc` single-threaded processes."""
    import multiprocessing as mp

    parts = np.array_split(np.arange(E_cols.shape[1]), nproc)
    tasks = [(np.ascontiguousarray(E_cols[:, p]), obs_ens, obs, perturbs, decorr, taper_rows[p]) for p in parts if len(p)]
    with mp.get_context("spawn").Pool(len(tasks)) as pool:  # spawn: the parent may hold a HIP context
        res = pool.map(_oracle_loc_task, tasks, chunksize=1)
    return np.concatenate(res, axis=1)


def fp32_vs_fp64_drift(grid, N, steps=40, every=1, seed=1, sat_variant32=0, dt=0.025, nTime=40, report=None):
    """dtype=32 plan against dtype=64 plan on the same inputs, advanced side by side without history (GPU).  After every `every`-th
    time step: max, 99.9-percentile and mean |S32 - S64| over all members and cells, the largest producer-series difference, the
    water-in-place difference per member (mean saturation = water in place / pore volume; max and min over members) and whether both
    modes took the same sub-step counts.  Returns the list of rows (dicts); `report(row)` is called as they come."""
    from historymatching_amd.forward import ForwardPlan

    _, g64 = make_models(grid, grid, dtype=64)
    _, g32 = make_models(grid, grid, dtype=32)
    x = perms(grid, grid, N, seed=seed)
    p64 = ForwardPlan(g64, N, dt, nTime, keep_history=False)
    p32 = ForwardPlan(g32, N, dt, nTime, keep_history=False)
    p32.set_variant(0, sat_variant32)
    p64.set_inputs(x, None, transformed=False)
    p32.set_inputs(x, None, transformed=False)
    rows = []
    try:
        for k in range(steps):
            p64.run(k, 1)
            p32.run(k, 1)
            if (k + 1) % every and k + 1 != steps:
                continue
            S64 = p64.get_field("S").reshape(N, -1)
            S32 = p32.get_field("S").reshape(N, -1).astype(np.float64)
            d = np.abs(S32 - S64)
            wip = S64.mean(axis=1) - S32.mean(axis=1)
            _, pr64, st64 = p64.outputs(want_wsats=False)
            _, pr32, st32 = p32.outputs(want_wsats=False)
            flat = d.reshape(-1)
            kth = int(0.999 * (flat.size - 1))
            row = dict(step=k + 1, max=float(d.max()), p999=float(np.partition(flat, kth)[kth]), mean=float(d.mean()),
                       prod=float(np.abs(pr32[:, :k + 1].astype(np.float64) - pr64[:, :k + 1]).max()), wip_max=float(wip.max()), wip_min=float(wip.min()),
                       same_nts=bool(np.array_equal(p64.get_field("nts")[:, :k + 1], p32.get_field("nts")[:, :k + 1])),
                       status=int(max(st64.max(), st32.max())), s_max=float(S32.max()), s_min=float(S32.min()))
            rows.append(row)
            if report:
                report(row)
    finally:
        p64.close()
        p32.close()
    return rows
