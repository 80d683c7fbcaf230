"""BASELINE.json configurations 3, 4 and 5 at their own sizes, against the oracle (-m gpu, through the C ABI).

  config 3: N_e = 1000, 128 x 128, fp32 matrix-core Kalman update, the reference's correlated observation error
  config 4: N_e = 4096, 256 x 256 over 8 GPUs  -> one rank's shard: 512 members
  config 5: N_e = 1000, 512 x 512, localised update (tapered covariance), fp32, over 8 GPUs -> shards of 125 members

What the oracle finishes in seconds is compared directly (a sample of state columns for the localised update, one time step
of two members at 512 x 512); the rest goes through size-independent properties."""
import numpy as np
import pytest

from tests.helpers import make_models, oracle_sims_and_noise_parallel, oracle_update_loc_columns, perms

pytestmark = pytest.mark.gpu

DT = 0.025


def _hm_inputs(N, M, n_obs, seed, nTime=40):
    """History-matching-shaped inputs with the reference's observation-error model (HistoryMatch.py:243-267, 638-639):
    R = kron(1e-2 toeplitz(exp(-k/2), cut below 1e-2), I_4), R12 its Cholesky factor, decorr = inv(R12^T), perturbations
    randn @ R12^T; the simulated observations depend linearly on a few state elements plus noise."""
    from oracle import es

    rng = np.random.RandomState(seed)
    E = (rng.randn(N, M) + rng.randn(M)).astype(np.float32).astype(np.float64)  # exactly representable in the fp32 plan
    _, R12, decorr = es.obs_error_model(nTime, n_obs // nTime)
    cols = rng.choice(M, n_obs, replace=False)
    obs_ens = 0.3 * E[:, cols] + 0.1 * rng.randn(N, n_obs)
    obs = obs_ens[0] + R12 @ rng.randn(n_obs)
    perturbs = rng.randn(N, n_obs) @ R12.T
    return E, obs_ens, obs, perturbs, decorr


def test_config3_full_size_fp32_update_vs_oracle():
    """N = 1000, M = 128 * 128, n_obs = 160, fp32 state contractions on the matrix cores, the correlated R of
    HistoryMatch.py:243-259: every element of the posterior against oracle.es.ens_update0 (fp64, the reference's
    association and pinv), bar 1e-4 of the largest increment."""
    from historymatching_amd.update import UpdatePlan, ens_update0
    from oracle import es

    N, M, n_obs = 1000, 128 * 128, 160
    E, obs_ens, obs, perturbs, decorr = _hm_inputs(N, M, n_obs, seed=33)
    ref = es.ens_update0(E, obs_ens, obs, perturbs, decorr)
    inc = np.abs(ref - E).max()
    assert inc > 0.1
    out32 = ens_update0(E, obs_ens, obs, perturbs, decorr, dtype=32)
    assert out32.dtype == np.float32 and np.abs(out32 - ref).max() <= 1e-4 * inc
    # the device-resident plan bench.py times (hm_upd_run: LDS-staged contractions, matrix-core inverse) is the same arithmetic
    plan = UpdatePlan(N, N, M, n_obs, dtype=32)
    plan.set_inputs(E, obs_ens, obs, perturbs, decorr)
    plan.run_local()
    out_plan = plan.output()
    plan.close()
    assert np.abs(out_plan - ref).max() <= 1e-4 * inc
    out64 = ens_update0(E, obs_ens, obs, perturbs, decorr, dtype=64)
    assert np.abs(out64 - ref).max() < 1e-10


def test_config3_one_full_size_es_mda_pass_vs_oracle():
    """Config 3 at its own size, one assimilation pass of the device-resident driver: N = 1000 members at 128 x 128 through
    `update.es_mda_device` in the arithmetic bench.py's `es_mda_config3` times (dtype=32 forward model -- fp32 saturation sweep, fp64
    pressure -- on the ensemble held by the update plan, producer series handed over on the device, fp32 matrix-core analysis step)
    against oracle.es.ens_update0 applied to the same prior with the GPU's own simulated observations and the same perturbations (HistoryMatch.py:578-586 with R12 -> sqrt(alpha) R12, alpha = n_iter = 1);
    every element, bar 1e-4 of the largest increment."""
    import scipy.linalg as sla

    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.update import es_mda_device
    from oracle import es

    n, N, nTime = 128, 1000, 40
    _, gm = make_models(n, n, dtype=32)
    prior = perms(n, n, N, seed=21).astype(np.float32).astype(np.float64)  # exactly representable in the fp32 plan
    _, R12, decorr = es.obs_error_model(nTime, 4)
    fwd = ForwardPlan(gm, N, DT, nTime, keep_history=False)
    fwd.set_inputs(prior, transformed=False)
    fwd.run()
    fwd.sync()
    _, prods, status = fwd.outputs(want_wsats=False)
    fwd.close()
    assert not status.any() and prods.dtype == np.float32
    obs_ens = es.vect(prods.astype(np.float64), nTime)
    obs = np.clip(obs_ens[0] + R12 @ np.random.RandomState(5).randn(4 * nTime), 0, 1)
    post = es_mda_device(gm, prior, obs, R12, DT, nTime, n_iter=1, rng=np.random.RandomState(77), dtype=32)
    perturbs = np.random.RandomState(77).randn(N, 4 * nTime) @ R12.T
    ref = es.ens_update0(prior, obs_ens, obs, perturbs, sla.inv(R12.T))
    inc = np.abs(ref - prior).max()
    assert inc > 0.05
    assert np.abs(post - ref).max() <= 1e-4 * inc


def test_config4_analysis_shape_whole_and_row_sharded_vs_oracle():
    """Config 4's analysis step at its own size: N = 4096 members, M = 256 * 256 state elements, n_obs = 160, fp32 plans, the
    reference's correlated R -- once as one plan holding every member (what bench.py times under `es_update.by_shape`) and once
    as the 8 row shards of 512 members the node's GPUs hold, run one after the other on this GPU through the three phases the
    ranks run, their reduce buffers summed on the host where the ranks all-reduce.  Both against
    oracle.es.ens_update0 on 2 048 random state columns (the update is column-separable in M: HistoryMatch.py:586 multiplies
    X from the left only), all 4096 members, bar 1e-4 of the largest increment."""
    from historymatching_amd.update import UpdatePlan
    from oracle import es

    N, M, G, n_obs = 4096, 256 * 256, 8, 160
    E, obs_ens, obs, perturbs, decorr = _hm_inputs(N, M, n_obs, seed=44)
    cols = np.sort(np.random.RandomState(6).choice(M, 2048, replace=False))
    ref = es.ens_update0(np.ascontiguousarray(E[:, cols]), obs_ens, obs, perturbs, decorr)
    inc = np.abs(ref - E[:, cols]).max()
    assert inc > 0.05
    plan = UpdatePlan(N, N, M, n_obs, dtype=32)
    plan.set_inputs(E, obs_ens, obs, perturbs, decorr)
    plan.run_local()
    whole = plan.output()[:, cols]
    plan.close()
    assert np.abs(whole - ref).max() <= 1e-4 * inc
    Nl = N // G
    plans = []
    for r in range(G):
        sl = slice(r * Nl, (r + 1) * Nl)
        p = UpdatePlan(N, Nl, M, n_obs, dtype=32)
        p.set_inputs(E[sl], obs_ens[sl], obs, perturbs[sl], decorr)
        plans.append(p)
    for ph in range(3):
        for p in plans:
            p.phase(ph)
        if ph < 2:
            for which in UpdatePlan.REDUCE_AFTER_PHASE[ph]:
                tot = sum(p.get_reduce(which).astype(np.float64) for p in plans)
                for p in plans:
                    p.set_reduce(which, tot)
    out = np.concatenate([(p.sync(), p.output()[:, cols])[1] for p in plans])
    for p in plans:
        p.close()
    assert np.abs(out - ref).max() <= 1e-4 * inc


def test_config4_shard_whole_run_properties():
    """One rank's shard of config 4: 512 members at 256 x 256, all 40 steps, default kernels (nested-dissection pressure solve of the
    larger grids + slab sweep: 8 rounds of 64 teams).  Mass-balance bracket, bounds, monotone producer series, and a sub-ensemble
    from different team rounds run alone agrees BIT FOR BIT (the direct solver treats every member by itself)."""
    n, N, steps = 256, 512, 40
    _, gm = make_models(n, n)
    from historymatching_amd.forward import ForwardPlan

    x = perms(n, n, N, seed=4)
    plan = ForwardPlan(gm, N, DT, steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0 and st["mean_nts"] > 2000
    assert S_end.min() >= -1e-9 and S_end.max() <= 1 + 1e-9 and np.isfinite(prods).all()
    assert (np.diff(prods, axis=1) >= -1e-9).all()
    fw = lambda s: s * s / (s * s + (1 - s) * (1 - s))  # noqa: E731
    water = S_end.sum(1) * (gm.Lx / n) * (gm.Ly / n)
    injected = steps * DT * 1.0
    at_end = DT * 0.25 * fw(prods).sum((1, 2))
    at_start = DT * 0.25 * fw(np.concatenate([np.zeros((N, 1, 4)), prods[:, :-1]], 1)).sum((1, 2))
    assert (water >= injected - at_end - 1e-9).all() and (water <= injected - at_start + 1e-9).all()
    sub = [0, 63, 64, 300, 511]
    plan = ForwardPlan(gm, len(sub), DT, steps, keep_history=False)
    plan.set_inputs(x[sub], transformed=False)
    plan.run()
    plan.sync()
    S_sub, p_sub, st2 = plan.outputs()
    plan.close()
    assert not st2.any()
    assert np.array_equal(S_sub, S_end[sub]) and np.array_equal(p_sub, prods[sub])


def test_config4_grid_whole_run_vs_oracle():
    """256 x 256 (config 4's grid), two members, ALL 40 time steps (2 458 explicit sub-steps each) through the default kernels -- nested
    dissection of the 13-level tree with its dry-front reuse, slab sweep -- against the oracle: within the oracle's own solver noise at every
    stored step (second SuperLU ordering: MMD_AT_PLUS_A), the sub-step count the CFL of the injector cell gives, the producer gather.  The
    whole-run companion of the one-solve comparison in test_assembly_bitexact_and_pressure_within_solver_noise[(256, 0)]."""
    from historymatching_amd.forward import ForwardPlan

    n, N, steps = 256, 2, 40
    om, gm = make_models(n, n)
    x = perms(n, n, N, seed=31)
    plan = ForwardPlan(gm, N, DT, steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0 and st["nd_fallbacks"] == 0
    assert (nts == 2458).all()  # ceil(2457.6), SURVEY.md Appendix B
    for m, (ref, noise) in enumerate(oracle_sims_and_noise_parallel(n, n, x, DT, steps, permc2="MMD_AT_PLUS_A")):
        err = np.abs(w[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (m, err, noise)
        assert err < 1e-4
        assert np.array_equal(p[m], w[m][1:, om.xy2ind(*om.prd_xy.T)])


def test_config5_grid_forward_vs_oracle():
    """512 x 512 (config 5's grid): two members, one time step (9 831 explicit sub-steps), default kernels -- nested-dissection
    pressure solve (press_nd512.o: 15 levels, fronts of up to 49 tile rows), saturation sweep by teams of 16 tile workgroups -- against the
    oracle, within the oracle's own solver noise (second SuperLU ordering: MMD_AT_PLUS_A; NATURAL fills in too much here)."""
    from historymatching_amd.forward import ForwardPlan

    n, N = 512, 2
    om, gm = make_models(n, n)
    x = perms(n, n, N, seed=23)
    plan = ForwardPlan(gm, N, DT, 1)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0
    assert (nts[:, 0] == 9831).all()  # ceil(9830.4), SURVEY.md Appendix B: the injector cell sets the CFL limit
    for m, (ref, noise) in enumerate(oracle_sims_and_noise_parallel(n, n, x, DT, 1, permc2="MMD_AT_PLUS_A")):
        err = np.abs(w[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (m, err, noise)
        assert err < 1e-4
        assert np.array_equal(p[m], w[m][1:, om.xy2ind(*om.prd_xy.T)])


def test_config5_grid_fp64_five_steps_vs_oracle():
    """512 x 512 over FIVE time steps (49 155 explicit sub-steps; the water front has crossed the leaf and subtree boundaries of the
    15-level tree around the injector and dry-front results are being reused): one member, default fp64 kernels, against the oracle within
    its own solver noise (second SuperLU ordering MMD_AT_PLUS_A), every stored step; same sub-step counts."""
    from historymatching_amd.forward import ForwardPlan

    n, steps = 512, 5
    om, gm = make_models(n, n)
    x = perms(n, n, 1, seed=29)
    plan = ForwardPlan(gm, 1, DT, steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0 and st["nd_fallbacks"] == 0
    assert (nts[0, :steps] == 9831).all()
    (ref, noise), = oracle_sims_and_noise_parallel(n, n, x, DT, steps, permc2="MMD_AT_PLUS_A")
    err = np.abs(w[0] - ref).max(axis=1)
    assert err.max() <= 10 * noise + 1e-9, (err, noise)
    assert err.max() < 1e-4
    assert np.array_equal(p[0], w[0][1:, om.xy2ind(*om.prd_xy.T)])


def test_config5_grid_fp32_forward_vs_oracle():
    """512 x 512 in the mode config 5 runs in: dtype=32 plans (slab teams of 16 workgroups, sat32s.hip; fp64 nested-dissection pressure
    solve), two members, three time steps = 29 493 explicit sub-steps, against the fp64 oracle: <= 1e-3 on S (a plain
    float32 accumulator is at 4.6e-4 after 4 steps and beyond 1e-3 from step 9 here).  Same sub-step counts."""
    from historymatching_amd.forward import ForwardPlan

    n, N, steps = 512, 2, 3
    om, gm = make_models(n, n, dtype=32)
    x = perms(n, n, N, seed=23)
    plan = ForwardPlan(gm, N, DT, steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0 and w.dtype == np.float32
    assert (nts[:, :steps] == 9831).all()
    for m, (ref, noise) in enumerate(oracle_sims_and_noise_parallel(n, n, x, DT, steps, permc2="MMD_AT_PLUS_A")):
        err = np.abs(w[m] - ref).max()
        assert err < 1e-3, (m, err, noise)
        assert err < 1e-4  # observed: a few 1e-6
        assert abs(float(w[m][-1].astype(float).mean()) - float(ref[-1].mean())) < 1e-7  # water in place


@pytest.mark.parametrize("grid,N", [(128, 1000), (256, 512), (512, 125)])
def test_fp32_forward_mode_whole_run_within_bar_of_fp64_mode(grid, N):
    """The dtype=32 forward mode against the dtype=64 forward mode on the same inputs over all 40 steps, at the sizes the fp32 mode is
    used: config 3 (1000 x 128^2), a config-4 shard (512 x 256^2), a config-5 shard (125 x 512^2).  Every fourth step: max |S32 - S64|
    over all members and cells <= 1e-3 (the stated bar; observed 6e-5 / 4e-4 / 6e-4, before the compensated accumulator 5e-4 / 2.4e-2 /
    0.30), 99.9 % of the cells within 1e-4, producer series within 1e-3, water in place within 1e-6 of the pore volume, identical
    sub-step counts (fp64 in both modes)."""
    from tests.helpers import fp32_vs_fp64_drift

    rows = fp32_vs_fp64_drift(grid, N, steps=40, every=4)
    assert len(rows) == 10
    for r in rows:
        assert r["status"] == 0 and r["same_nts"], r
        assert r["max"] <= 1e-3 and r["p999"] <= 1e-4 and r["prod"] <= 1e-3, r
        assert max(abs(r["wip_max"]), abs(r["wip_min"])) <= 1e-6, r
        # (round 6: the injector's source term is rounded jointly with the cell's diagonal coefficient, sat32.h: source32 -- the cell fills up
        # to 1 and stays; before that it crept above 1 by up to 1.7e-4, one float32 ulp of fi d per sub-step)
        assert -1e-6 <= r["s_min"] and r["s_max"] <= 1 + 1e-6, r


@pytest.mark.parametrize("dtype", [64, 32])
def test_config5_grid_properties_over_steps(dtype):
    """512 x 512, 4 members, 3 steps: no producer has seen water yet, so the water in place is exactly the injected volume;
    bounds; members independent of their batch (same split of the CG passes: both batches have fewer members than CUs / 8)."""
    from historymatching_amd.forward import ForwardPlan

    n, N, steps = 512, 4, 3
    _, gm = make_models(n, n, dtype=dtype)
    x = perms(n, n, N, seed=6)
    plan = ForwardPlan(gm, N, DT, steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    tol = 1e-9 if dtype == 64 else 1e-6
    assert not status.any() and np.abs(prods).max() == 0
    S_end = S_end.astype(float)
    assert S_end.min() >= -tol and S_end.max() <= 1 + tol  # (fp32: S <= 1 at the injector's cell by construction, sat32.h: source32)
    water = S_end.sum(1) * (gm.Lx / n) * (gm.Ly / n)
    assert np.abs(water - steps * DT).max() < (1e-9 if dtype == 64 else 1e-6)  # fp32: the compensated pair of csrc/sat32.h (1e-4 before it)
    plan = ForwardPlan(gm, 2, DT, steps, keep_history=False)
    plan.set_inputs(x[[3, 1]], transformed=False)
    plan.run()
    plan.sync()
    S_sub, _, st2 = plan.outputs()
    plan.close()
    assert not st2.any() and np.abs(S_sub.astype(float) - S_end[[3, 1]]).max() < (1e-6 if dtype == 64 else 1e-4)


def test_config5_shard_whole_run_properties():
    """One rank's shard of config 5 as the 8-GPU configuration runs it: 125 members at 512 x 512, all 40 steps, fp32 plans (fp32 slab-team
    sweep sat32s.hip with the compensated state of sat32.h, fp64 nested-dissection pressure solve press_nd512.o) -- late-run behaviour
    included.  Mass-balance bracket, bounds, monotone producer series, and a sub-ensemble from different team rounds (16 members a
    round) run alone agrees bit for bit (the direct solver and the sweep treat every member by itself)."""
    from historymatching_amd.forward import ForwardPlan

    n, N, steps = 512, 125, 40
    _, gm = make_models(n, n, dtype=32)
    x = perms(n, n, N, seed=8)
    plan = ForwardPlan(gm, N, DT, steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    om, _ = make_models(n, n)
    q = om.source_field(0)[0]
    worst = 0.0
    for k0 in range(0, steps, 8):  # the direct solver's divergence residual, what its a-posteriori check sees, at steps 8, 16, ... 40
        plan.run(k0, 8)
        plan.sync()
        Vx, Vy = plan.get_field("Vx"), plan.get_field("Vy")
        for m0 in range(0, N, 25):
            div = (Vx[m0:m0 + 25, 1:] - Vx[m0:m0 + 25, :-1]) + (Vy[m0:m0 + 25, :, 1:] - Vy[m0:m0 + 25, :, :-1])
            worst = max(worst, float(np.abs(div.reshape(len(div), -1) - q).max()))
        del Vx, Vy
    st = plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0 and st["nd_fallbacks"] == 0 and st["mean_nts"] > 9000
    # the check's threshold is 1e-4 max |q| = 1e-4; a healthy solve leaves T eps |p| there (1e-13 ... 1e-5 over config 4's prior at
    # 256 x 256, profiles/r04/nd_residual_stats.txt): a factor 3 of margin over a whole run at this grid
    assert worst < 3e-5, worst
    S64, p64 = S_end.astype(float), prods.astype(float)
    tol = 1e-6
    # The injector's cell sits at S = 1 (fw = 1) and is stationary there when c_C + fi d = 0 in the arithmetic of the sweep.  Rounds 1-5
    # rounded the two to float32 independently: they then differ by an ulp that the cell gains every sub-step, 9 831 x 40 times -- up to
    # 1 + 1.7e-4 depending on the last bits of the fluxes (profiles/diag/c5_smax.py), and the bound here had to follow (1e-4, then 1e-3).
    # Round 6 rounds them JOINTLY (sat32.h: source32, fid32 = fl32((c_C + fi d) - cC32)): the sum is exact, S <= 1 by construction.
    assert S64.min() >= -tol and S64.max() <= 1 + tol and np.isfinite(p64).all()
    inj_cell = gm.xy2ind(*np.asarray(gm.inj_xy).T)
    assert (S64[:, inj_cell] <= 1.0).all() and (S64[:, inj_cell] > 0.99).all(), (S64[:, inj_cell].min(), S64[:, inj_cell].max())
    assert (np.diff(p64, axis=1) >= -tol).all()
    fw = lambda s: s * s / (s * s + (1 - s) * (1 - s))  # noqa: E731
    water = S64.sum(1) * (gm.Lx / n) * (gm.Ly / n)
    injected = steps * DT * 1.0
    at_end = DT * 0.25 * fw(p64).sum((1, 2))
    at_start = DT * 0.25 * fw(np.concatenate([np.zeros((N, 1, 4)), p64[:, :-1]], 1)).sum((1, 2))
    # (fp32 state over 40 x 9 831 sub-steps: with the compensated pair the bracket holds to 1e-5 of the injected volume of 1.0 -- the
    # plain float32 accumulator of rounds 1-4 needed 5e-3 here; 1e-9 in fp64)
    viol = max(float((injected - at_end - water).max()), float((water - (injected - at_start)).max()))
    assert viol < 1e-5, viol
    sub = [0, 15, 16, 77, 124]
    plan = ForwardPlan(gm, len(sub), DT, steps, keep_history=False)
    plan.set_inputs(x[sub], transformed=False)
    plan.run()
    plan.sync()
    S_sub, p_sub, st2 = plan.outputs()
    plan.close()
    assert not st2.any()
    assert np.array_equal(S_sub, S_end[sub]) and np.array_equal(p_sub, prods[sub])


def test_config5_localised_update_at_shard_shape_vs_oracle():
    """Config 5's analysis step at its own size: N = 1000 members in 8 row shards of 125 (one per GPU of the node), M = 512 * 512
    state elements, n_obs = 160, fp32 plans (matrix-core contractions and local analyses), taper = bump(dist / 1.2) from the
    product's `taper_for_wells` on the 512 x 512 grid with the reference wells (HistoryMatch.py:700-717, 863).  The eight
    shards run one after the other on this GPU through the same three phases the ranks run, their reduce buffers summed
    on the host where the ranks all-reduce.  Compared with oracle.es.ens_update0_loc on 2 048 random state columns (all
    1000 members), bar 1e-4 of the largest increment."""
    from historymatching_amd.localization import taper_for_wells
    from historymatching_amd.update import UpdatePlan

    n, N, G, n_obs = 512, 1000, 8, 160
    M = n * n
    _, gm = make_models(n, n)
    taper = taper_for_wells(gm, gm.xy2ind(*gm.prd_xy.T), 40, radius=1.2)
    assert taper.shape == (M, n_obs) and (taper > 0).any(1).mean() > 0.9
    E, obs_ens, obs, perturbs, decorr = _hm_inputs(N, M, n_obs, seed=55)
    Nl = N // G
    plans = []
    for r in range(G):
        sl = slice(r * Nl, (r + 1) * Nl)
        p = UpdatePlan(N, Nl, M, n_obs, dtype=32, localized=True)
        p.set_inputs(E[sl], obs_ens[sl], obs, perturbs[sl], decorr, taper)
        plans.append(p)
    for ph in range(3):
        for p in plans:
            p.phase(ph)
        if ph < 2:
            for which in UpdatePlan.REDUCE_AFTER_PHASE[ph]:
                tot = sum(p.get_reduce(which).astype(np.float64) for p in plans)
                for p in plans:
                    p.set_reduce(which, tot)
    cols = np.sort(np.random.RandomState(3).choice(M, 2048, replace=False))
    out = np.concatenate([(p.sync(), p.output()[:, cols])[1] for p in plans])
    for p in plans:
        p.close()
    ref = oracle_update_loc_columns(E[:, cols], obs_ens, obs, perturbs, decorr, taper[cols])
    inc = np.abs(ref - E[:, cols]).max()
    assert inc > 0.05
    n_loc = (np.sqrt(taper[cols]) > 1e-2).sum(1)
    assert n_loc.max() == n_obs and n_loc.min() < n_obs  # the sample spans full and partial local domains
    assert np.abs(out - ref).max() <= 1e-4 * inc


def test_config5_localised_es_mda_pass_wiring():
    """One localised ES-MDA pass end to end at 512 x 512 (16 members, all 40 steps, fp32 plans): `dist.es_mda_sharded` on one
    rank -- forward model, producer series handed to the update plan on the device, localised analysis, posterior fetched --
    equals oracle.es.ens_update0_loc applied to the same prior with the GPU's own simulated observations (sampled columns)."""
    from historymatching_amd.dist import es_mda_sharded
    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.localization import taper_for_wells
    from oracle import es

    n, N, nTime = 512, 16, 40
    _, gm = make_models(n, n, dtype=32)
    prior = perms(n, n, N, seed=8).astype(np.float32).astype(np.float64)
    taper = taper_for_wells(gm, gm.xy2ind(*gm.prd_xy.T), nTime, radius=1.2)
    _, R12, decorr = es.obs_error_model(nTime, 4)
    fwd = ForwardPlan(gm, N, DT, nTime, keep_history=False)
    fwd.set_inputs(prior, transformed=False)
    fwd.run()
    fwd.sync()
    _, prods, status = fwd.outputs(want_wsats=False)
    fwd.close()
    assert not status.any()
    obs_ens = es.vect(prods.astype(np.float64), nTime)
    obs = np.clip(obs_ens[0] + R12 @ np.random.RandomState(2).randn(4 * nTime), 0, 1)
    post = es_mda_sharded(gm, prior, obs, R12, DT, nTime, n_iter=1, seed=11, dtype=32, taper=taper)
    perturbs = np.random.RandomState(11).randn(N, 4 * nTime) @ R12.T
    cols = np.sort(np.random.RandomState(4).choice(n * n, 256, replace=False))
    ref = oracle_update_loc_columns(prior[:, cols], obs_ens, obs, perturbs, decorr, taper[cols], nproc=4)
    inc = np.abs(ref - prior[:, cols]).max()
    assert np.abs(post[:, cols] - ref).max() <= 1e-4 * max(inc, 1e-3)
