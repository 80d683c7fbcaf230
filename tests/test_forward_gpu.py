"""GPU parity tests of the ensemble forward model (pressure + saturation) against oracle/ressim.py.

Bars (DESIGN.md "Parity"):
  * transmissibility assembly, CFL sub-step count and the saturation sweep: BIT-EXACT in fp64
    (same operations in the same order as the NumPy/SciPy restatement);
  * pressure solve: different direct solver than the oracle's SuperLU, and the system is ill-conditioned
    (T up to ~1e6 * eps * |p| ~ 1e-9 absolute flux noise for ANY fp64 solver), so the bar is a tolerance
    tied to the oracle's own ordering noise (COLAMD vs NATURAL SuperLU): |V_gpu - V_oracle| <= 10 x that + 1e-12;
  * whole simulation: 10 x the oracle's own ordering noise + 1e-9 abs on S, per member.
"""
import numpy as np
import pytest

from tests.helpers import make_models, oracle_sim_and_noise, perms

pytestmark = pytest.mark.gpu

DT, NT = 0.025, 40


def _plan(gm, N, nTime=NT, keep_history=True):
    from historymatching_amd.forward import ForwardPlan

    return ForwardPlan(gm, N, DT, nTime, keep_history=keep_history)


def _oracle_state(om, x, nsteps):
    """Advance the oracle `nsteps` steps; return (S, q) at that point."""
    from oracle.ressim import set_perm

    set_perm(om, x)
    S = np.zeros(om.Nxy)
    q, _, _ = om.source_field(0)
    for _ in range(nsteps):
        _, Vx, Vy = om.pressure_step(S, q)
        S = om.saturation_step_upwind(S, q, Vx, Vy, DT)
    return S, q


def test_device_is_gfx950():
    from historymatching_amd import _lib

    name = _lib.Context.get().name()
    assert "gfx950" in name, name


def test_mfma_f64_layout():
    """Operand / result lane maps of v_mfma_f64_16x16x4_f64 assumed by press128m.hip (asymmetric integer data)."""
    import ctypes as C

    from historymatching_amd import _lib

    ctx = _lib.Context.get()
    A = np.arange(64, dtype=np.float64).reshape(16, 4) + 1
    B = (np.arange(64, dtype=np.float64).reshape(4, 16) * 3 - 50)
    D = np.zeros((16, 16))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    _lib.check(ctx.lib.hm_debug_mfma_f64(ctx.handle, dp(A), dp(B), dp(D)), "hm_debug_mfma_f64")
    assert np.array_equal(D, A @ B)


def test_float32_fractional_flow_is_the_ieee_quotient_for_every_operand():
    """dtype = 32 plans compute fw(s) = s^2 / (s^2 + (1 - s)^2) with a four-instruction division (csrc/fracflow.h).  fw has ONE float operand:
    the kernels' function is compared with the IEEE division on all 2^32 of them, on the GPU the tests run on (the reciprocal seed is the
    hardware's).  This is what makes the fp32 sweeps equal to the NumPy float32 specification (oracle/ressim.py) bit for bit."""
    import ctypes as C

    from historymatching_amd import _lib

    ctx = _lib.Context.get()
    out = (C.c_ulonglong * 2)()
    _lib.check(ctx.lib.hm_debug_fracflow32_check(ctx.handle, out), "hm_debug_fracflow32_check")
    assert out[0] == 0, f"{out[0]} float32 operands below 2^62 give a fractional flow that is not the IEEE quotient"
    assert out[1] < 2**25  # (operands beyond 6.5e18: s^2 near the top of the float32 range; no saturation is ever there)


def test_fp64_fractional_flow_is_the_ieee_quotient_on_a_dense_operand_set():
    """The fp64 sweeps divide in seven instructions (csrc/fracflow.h: div_unscaled -- one cubic refinement of the reciprocal instead of two
    quadratic ones).  A double has too many bit patterns to try them all as the float32 form is; the claim "the IEEE quotient bit for bit" is pinned
    on 1.7e10 operands where saturations live: dense over [0, 1 + 2^-9), clustered next to 0, 1/2 and 1, every binade down to the denormals
    (hm_debug_fracflow64_check).  Zero differences from the compiler's IEEE division."""
    import ctypes as C

    from historymatching_amd import _lib

    ctx = _lib.Context.get()
    out = (C.c_ulonglong * 2)()
    _lib.check(ctx.lib.hm_debug_fracflow64_check(ctx.handle, out), "hm_debug_fracflow64_check")
    assert out[1] >= 2**34 and out[0] == 0, f"{out[0]} of {out[1]} fp64 operands give a fractional flow that is not the IEEE quotient"


def test_perm_transform_on_device():
    om, gm = make_models(20, 20)
    x = perms(20, 20, 5)
    plan = _plan(gm, 5, nTime=1)
    plan.set_inputs(x, transformed=False)
    K = plan.get_field("K").reshape(5, -1)
    ref = 0.1 + np.exp(5 * x)
    # exp() of two correctly-rounding-within-1ulp libraries may differ in the last place
    assert np.max(np.abs(K - ref) / ref) < 4e-16


@pytest.mark.parametrize("n,variant", [(20, 1), (20, 0), (20, 9), (128, 1), (128, 7), (128, 9), (128, 0), (128, 12), (128, 13),
                                       (160, 0), (256, 0), (256, 15)])
def test_assembly_bitexact_and_pressure_within_solver_noise(n, variant):
    """(256, 0): the nested dissection of the larger grids (press_nd256.o: big-front kernels); (256, 15): the two-level CG there."""
    from oracle.ressim import perm_transf
    from scipy.sparse.linalg import spsolve

    om, gm = make_models(n, n)
    N = 3
    x = perms(n, n, N, seed=3)
    plan = _plan(gm, N, nTime=2)
    plan.set_variant(variant, variant)
    plan.set_inputs(perm_transf(x), transformed=True)
    S_all, V_all = [], []
    for m in range(N):
        S, q = _oracle_state(om, x[m], 2 if n == 20 else 1)
        S_all.append(S)
    plan.set_field("S", np.array(S_all))
    plan.pressure_only(0)
    TX, TY = plan.get_field("TX"), plan.get_field("TY")
    P, Vx, Vy = plan.get_field("P"), plan.get_field("Vx"), plan.get_field("Vy")
    for m in range(N):
        from oracle.ressim import set_perm

        set_perm(om, x[m])
        Mw, Mo = om.rel_perm(S_all[m])
        KM = (Mw + Mo).reshape(om.shape) * om.K
        L = KM ** (-1)
        TXo = np.zeros((n + 1, n))
        TYo = np.zeros((n, n + 1))
        TXo[1:-1, :] = 2 * om.hy / om.hx / (L[0, :-1, :] + L[0, 1:, :])
        TYo[:, 1:-1] = 2 * om.hx / om.hy / (L[1, :, :-1] + L[1, :, 1:])
        assert np.array_equal(TX[m], TXo), "x-transmissibilities must be bit-exact"
        assert np.array_equal(TY[m], TYo), "y-transmissibilities must be bit-exact"
        # oracle solve (COLAMD) and a second ordering to measure the reference's own noise floor
        Po, Vxo, Vyo = om.pressure_step(S_all[m], q)
        x1, x2 = TXo[:-1].ravel(), TXo[1:].ravel()
        y1, y2 = TYo[:, :-1].ravel(), TYo[:, 1:].ravel()
        diag = y1 + y2 + x1 + x2
        diag[0] += np.sum(om.K[:, 0, 0])
        A = om.spdiags([-x2, -y2, diag, -y1, -x1], [-n, -1, 0, 1, n]).tocsc()
        Pn = spsolve(A, q, permc_spec="NATURAL").reshape(n, n)
        Vxn = np.zeros_like(Vxo)
        Vxn[1:-1] = (Pn[:-1] - Pn[1:]) * TXo[1:-1]
        noise = max(np.abs(Vxn - Vxo).max(), 1e-15)
        err = max(np.abs(Vx[m] - Vxo).max(), np.abs(Vy[m] - Vyo).max())
        assert err <= 10 * noise + 1e-12, (err, noise)
        # discrete divergence of the GPU fluxes reproduces the wells (mass conservation)
        div = (Vx[m][1:] - Vx[m][:-1]) + (Vy[m][:, 1:] - Vy[m][:, :-1])
        assert np.abs(div.ravel() - q).max() <= 10 * noise + 1e-10
        assert np.abs(P[m] - Po).max() <= 1e-5 * np.abs(Po).max()
    plan.close()


@pytest.mark.parametrize("n,variant", [(20, 1), (20, 0), (128, 1), (128, 0)])
def test_saturation_step_bitexact_given_fluxes(n, variant):
    """Same V in, same S out, to the last bit, including the CFL sub-step count."""
    from oracle.ressim import perm_transf, set_perm

    om, gm = make_models(n, n)
    N = 3
    x = perms(n, n, N, seed=5)
    plan = _plan(gm, N, nTime=2)
    plan.set_variant(variant, variant)
    plan.set_inputs(perm_transf(x), transformed=True)
    S_in, Vxs, Vys, S_ref, nts_ref = [], [], [], [], []
    for m in range(N):
        S, q = _oracle_state(om, x[m], 2 if n == 20 else 1)
        set_perm(om, x[m])
        _, Vx, Vy = om.pressure_step(S, q)
        S_in.append(S), Vxs.append(Vx), Vys.append(Vy)
        S_ref.append(om.saturation_step_upwind(S, q, Vx, Vy, DT))
        nts_ref.append(om.cfl_substeps(Vx, Vy, q, DT)[0])
    plan.set_field("S", np.array(S_in))
    plan.set_field("Vx", np.array(Vxs))
    plan.set_field("Vy", np.array(Vys))
    plan.saturation_only(0)
    S_gpu = plan.get_field("S").reshape(N, -1)
    nts = plan.get_field("nts")[:, 0]
    assert nts.tolist() == nts_ref
    assert nts_ref[0] == (15 if n == 20 else 615)  # ceil(14.999999999999998), ceil(614.4)
    for m in range(N):
        assert np.array_equal(S_gpu[m], S_ref[m]), np.abs(S_gpu[m] - S_ref[m]).max()
    plan.close()


def _adversarial_saturations(n, dtype, seed):
    """Saturation fields made of the values the division of the fractional flow treats specially (csrc/fracflow.h): exact and
    negative zero, denormals, tiny normals on both sides of the thresholds of the compiler's operand scaling (S^2 around 2^-970 /
    2^-103), values whose square underflows, the ends of [0, 1] and a little outside, next to ordinary saturations."""
    rng = np.random.RandomState(seed)
    ft = np.float64 if dtype == 64 else np.float32
    fi = np.finfo(ft)
    emin = fi.minexp  # -1022 / -126
    special = [0.0, -0.0, fi.smallest_subnormal, 3 * fi.smallest_subnormal, fi.tiny, fi.tiny * (1 + fi.eps), 1.0, 1.0 - fi.epsneg,
               1.0 + fi.eps, 0.5, 1.2, -fi.tiny, -1e-17 if dtype == 64 else -1e-8, 1e-30, 1e-3]
    S = rng.rand(n * n).astype(ft)
    k = rng.rand(n * n)
    # a third of the cells: mantissa x 2^e with e spread over the whole tiny range, denser around the thresholds
    e_all = rng.randint(emin - 40, -2, n * n)
    e_thr = np.array([emin // 2 - 27, emin // 2 - 1, emin // 2, emin // 2 + 1, -485, -484, -481, -480, -479, -52, -51, -27, -26, -25, -24])
    e_near = e_thr[rng.randint(0, len(e_thr), n * n)] + rng.randint(-1, 2, n * n)
    tiny = np.ldexp(1.0 + rng.rand(n * n), np.where(rng.rand(n * n) < 0.5, e_all, e_near)).astype(ft)
    S = np.where(k < 0.35, tiny, S)
    S = np.where((k >= 0.35) & (k < 0.5), np.array(special, dtype=ft)[rng.randint(0, len(special), n * n)], S)
    S = np.where((k >= 0.5) & (k < 0.55), -tiny, S)
    return S.astype(ft)


def test_fractional_flow_division_bitexact_on_adversarial_saturations():
    """The sweeps divide without the compiler's operand scaling and fix-up instructions (csrc/fracflow.h).  One time step from
    saturation fields built of the values where those instructions act -- zeros, denormals, squares that underflow, tiny normals
    around the scaling thresholds, the ends of [0, 1] -- equals the NumPy restatement to the last bit, and the generic kernel
    (compiler's division)."""
    from oracle.ressim import perm_transf, set_perm

    n, N = 128, 4
    om, gm = make_models(n, n)
    x = perms(n, n, N, seed=77)
    S_in, Vxs, Vys, S_ref = [], [], [], []
    for m in range(N):
        S_real, q = _oracle_state(om, x[m], 1)
        set_perm(om, x[m])
        _, Vx, Vy = om.pressure_step(S_real, q)
        S = _adversarial_saturations(n, 64, 100 + m)
        if m == 3:
            S[: n * n // 2] = 0.0  # dry bands next to adversarial ones
        S_in.append(S), Vxs.append(Vx), Vys.append(Vy)
        with np.errstate(all="ignore"):
            S_ref.append(om.saturation_step_upwind(S, q, Vx, Vy, DT))
    out = {}
    for variant in (0, 1):
        plan = _plan(gm, N, nTime=2)
        plan.set_variant(variant, variant)
        plan.set_inputs(perm_transf(x), transformed=True)
        plan.set_field("S", np.array(S_in))
        plan.set_field("Vx", np.array(Vxs))
        plan.set_field("Vy", np.array(Vys))
        plan.saturation_only(0)
        out[variant] = plan.get_field("S").reshape(N, -1)
        plan.close()
    for m in range(N):
        assert np.array_equal(out[0][m], S_ref[m]), (m, np.abs(out[0][m] - S_ref[m]).max())
        assert np.array_equal(out[1][m], S_ref[m])
        assert np.array_equal(np.signbit(out[0][m]), np.signbit(out[1][m]))


def test_fractional_flow_division_fp32_bitexact_on_adversarial_saturations():
    """The same for dtype=32 plans: the register-resident fp32 sweep (unscaled single-precision division) against the generic fp32
    kernel (compiler's division) on adversarial fp32 saturations, same fluxes."""
    n, N = 128, 3
    om, gm = make_models(n, n, dtype=32)
    x = perms(n, n, N, seed=78)
    out = {}
    S_in = np.array([_adversarial_saturations(n, 32, 200 + m) for m in range(N)])
    Vx = Vy = None
    for variant in (1, 0):
        plan = _plan(gm, N, nTime=2)
        plan.set_variant(1, variant)
        plan.set_inputs(x, transformed=False)
        if Vx is None:
            plan.run(0, 1)  # a realistic flux field (one step from S = 0), reused for both kernels
            plan.sync()
            Vx, Vy = plan.get_field("Vx"), plan.get_field("Vy")
            plan.set_inputs(x, transformed=False)
        plan.set_field("S", S_in)
        plan.set_field("Vx", Vx)
        plan.set_field("Vy", Vy)
        plan.saturation_only(0)
        out[variant] = (plan.get_field("S").copy(), plan.get_field("nts")[:, 0].copy())
        plan.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0].view(np.uint32) & 0x7FFFFFFF, out[1][0].view(np.uint32) & 0x7FFFFFFF) or np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("nx,ny,general_fluid", [(128, 128, False), (128, 128, True), (256, 256, False), (256, 128, False), (128, 256, False), (96, 80, False)])
def test_fp32_sweeps_equal_the_float32_specification_bit_for_bit(nx, ny, general_fluid):
    """dtype=32 plans carry the saturation as a compensated float32 pair (csrc/sat32.h).  One time step of every fp32 sweep -- the
    register sweep on slabs (sat32s.hip; 128 x 128: one workgroup; 256 x 256, 256 x 128, 128 x 256: teams of 4 / 2 / 2 slabs trading
    rows as granules), the generic, streaming and tiled kernels -- from a mid-run state and its own fluxes equals
    oracle/ressim.py:saturation_step_stencil_f32c (NumPy float32, operation for operation) to the last bit: realistic members, one
    member of adversarial values (zeros of both signs, denormals, squares that underflow; half of it dry)."""
    from oracle.ressim import set_perm

    N, pre = 3, 2
    om, gm = make_models(nx, ny, dtype=32)
    if general_fluid:
        for mdl in (om, gm):
            mdl.vw, mdl.vo, mdl.swc, mdl.sor = 0.7, 1.9, 0.05, 0.1
    x = perms(nx, ny, N, seed=83)
    q, _, _ = om.source_field(0)
    ref = None
    for sat_variant in (0, 1, 2, 3):
        plan = _plan(gm, N, nTime=pre + 1, keep_history=False)
        plan.set_variant(0, sat_variant)
        w0 = None if not general_fluid else np.full((N, nx * ny), om.swc, dtype=np.float32)
        plan.set_inputs(x, w0, transformed=False)
        plan.run(0, pre)
        S = plan.get_field("S").reshape(N, -1)
        if not general_fluid and nx == ny:
            S[2] = _adversarial_saturations(nx, 32, 300)
            S[2, : nx * ny // 2] = 0.0
            plan.set_field("S", S)
        plan.pressure_only(pre)
        Vx, Vy = plan.get_field("Vx"), plan.get_field("Vy")
        plan.saturation_only(pre)
        out = plan.get_field("S").reshape(N, -1)
        nts = plan.get_field("nts")[:, pre]
        _, _, status = plan.outputs(want_wsats=False)
        plan.close()
        assert not status.any()
        # the specification on the first kernel's state and fluxes; every kernel reaches the same state bit for bit -- except where the
        # default variant runs embedded in a 128 x 128 plan (96 x 80: another pressure solver than the generic pair the other variants
        # keep on such a grid, so another state at rounding level): there the specification is evaluated again on the new state
        if ref is None or (max(nx, ny) < 128 and sat_variant == 1):
            ref = []
            for m in range(N):
                set_perm(om, x[m])
                om._trace = []
                with np.errstate(all="ignore"):
                    ref.append(om.saturation_step_stencil_f32c(S[m], q, Vx[m], Vy[m], DT))
                assert om._trace[0][0] == nts[m]
            S_first = S.copy()
        assert np.array_equal(S.view(np.uint32), S_first.view(np.uint32))  # same input state from every kernel
        for m in range(N):
            assert np.array_equal(out[m], ref[m]), (sat_variant, m, np.abs(out[m] - ref[m]).max())
    assert out[:2].max() > 0.5


def test_full_sim_20x20_matches_oracle():
    """C1-shaped case (reference default grid, HistoryMatch.py:97,219-221): 40 steps, whole history."""
    from oracle.ressim import forward_model as oracle_forward

    om, gm = make_models(20, 20)
    N = 6
    x = perms(20, 20, N, seed=7)
    w_ref, p_ref = oracle_forward(om, x, None, DT, NT)
    from historymatching_amd.forward import make_forward_model

    fm = make_forward_model(gm, DT, NT)
    w, p = fm(x)
    assert w.shape == (N, NT + 1, 400) and p.shape == (N, NT, 4)
    assert np.array_equal(w[:, 0], np.zeros((N, 400)))  # row 0 == wsat0 (HistoryMatch.py:225)
    prod_inds = om.xy2ind(*om.prd_xy.T)
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[m], DT, NT)
        assert np.array_equal(ref, w_ref[m])
        assert np.abs(w[m] - ref).max() <= 10 * noise + 1e-9, (m, np.abs(w[m] - ref).max(), noise)
        assert np.array_equal(p[m], w[m][1:, prod_inds])  # obs_model gather (HistoryMatch.py:212-213, 363)
    assert np.abs(w - w_ref).max() < 1e-6 and np.abs(p - p_ref).max() < 1e-6
    assert w.min() >= 0 and w.max() <= 1  # monotone under the CFL limit


@pytest.mark.parametrize("n,N,nT", [(128, 64, 40), (256, 40, 16)])
def test_drop_in_call_streams_a_large_history_out_while_it_runs(n, N, nT):
    """forward_model(perms) -> [wsats, prods] (HistoryMatch.py:383-387) with host arrays in and out: from 256 MB of saturation
    history on, time index k of every member leaves the device while step k runs (hm_fwd_run_to_host, hm_d2h_rows).
    Bit-identical to the device-resident plan's history read back in one piece, row 0 == wsat0, with and without given start states;
    at 256 x 256 the steps themselves are synchronous for the host (the CG pressure solve polls convergence) and the rows go out in
    between."""
    from historymatching_amd.forward import make_forward_model

    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=23)
    rng = np.random.RandomState(5)
    w0 = rng.uniform(0.0, 0.2, (N, n * n))
    for start in ((None, w0) if n == 128 else (None,)):
        plan = _plan(gm, N, nTime=nT)
        plan.set_inputs(x, start, transformed=False)
        plan.run()
        plan.sync()
        w_ref, p_ref, status = plan.outputs()
        plan.close()
        assert not status.any()
        fm = make_forward_model(gm, DT, nT)
        w, p = fm(x) if start is None else fm(x, start)
        fm.release()
        assert w.shape == (N, nT + 1, n * n) and w.nbytes >= 256 << 20  # the streamed path
        assert np.array_equal(w, w_ref) and np.array_equal(p, p_ref)
        assert np.array_equal(w[:, 0], np.zeros_like(w0) if start is None else w0)


def test_large_ensemble_runs_as_two_member_blocks_with_identical_results():
    """forward_model on the 128 x 128 kernels splits an ensemble of 512 members or more into two blocks on two streams, one host
    thread each (forward.make_forward_model); members are independent, so histories, producer series and member order equal the
    one-block plan's bit for bit; the statistics cover both blocks."""
    from historymatching_amd.forward import make_forward_model

    n, N, nT = 128, 520, 3
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=29)
    w0 = np.random.RandomState(7).uniform(0.0, 0.2, (N, n * n))
    plan = _plan(gm, N, nTime=nT)
    plan.set_inputs(x, w0, transformed=False)
    plan.run()
    st_ref = plan.sync()
    w_ref, p_ref, status = plan.outputs()
    plan.close()
    assert not status.any()
    fm = make_forward_model(gm, DT, nT)
    for _ in range(2):  # the second call reuses both plans
        w, p = fm(x, w0)
        assert np.array_equal(w, w_ref) and np.array_equal(p, p_ref)
    assert gm.last_stats["member_steps"] == st_ref["member_steps"] == N * nT
    assert gm.last_stats["n_saturation_launches"] == 2 * nT and abs(gm.last_stats["mean_nts"] - st_ref["mean_nts"]) < 1.0
    fm.release()


def test_device_resident_ensemble_in_member_blocks_is_bit_identical():
    """forward.BlockedForwardPlan (what bench.py times, and the arrangement forward_model uses): an ensemble of 768 members or more on the
    128 x 128 kernels is three member blocks on three streams, launches interleaved step by step; uneven bounds work as well; outputs and
    member order equal the one-block plan's bit for bit, the statistics cover all blocks."""
    from historymatching_amd.forward import BlockedForwardPlan, default_blocks

    n, N, nT = 128, 770, 3
    _, gm = make_models(n, n)
    assert (default_blocks(gm, 1000), default_blocks(gm, 767), default_blocks(gm, 511)) == (3, 2, 1)
    x = perms(n, n, N, seed=31)
    plan = _plan(gm, N, nTime=nT)
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    st_ref = plan.sync()
    w_ref, p_ref, status = plan.outputs()
    plan.close()
    assert not status.any()
    for kw in (dict(), dict(bounds=[0, 100, 770])):
        bp = BlockedForwardPlan(gm, N, DT, nT, keep_history=True, **kw)
        assert len(bp.plans) == (3 if not kw else 2)
        bp.set_inputs(x, None, transformed=False)
        bp.run(0, 2)
        bp.run(2, 1)
        st = bp.sync()
        w, p, status = bp.outputs()
        bp.close()
        assert not status.any() and np.array_equal(w, w_ref) and np.array_equal(p, p_ref)
        assert st["member_steps"] == st_ref["member_steps"] == N * nT and st["n_saturation_launches"] == len(bp.plans) * nT


def test_restart_from_member_states_and_order():
    """forward_model(perms, wsat0s): two zipped ensembles, member order preserved (HistoryMatch.py:1224-1227)."""
    from oracle.ressim import forward_model as oracle_forward
    from historymatching_amd.forward import make_forward_model

    om, gm = make_models(20, 20)
    N = 4
    x = perms(20, 20, N, seed=11)
    fm = make_forward_model(gm, DT, 10)
    w1, _ = fm(x)
    w2, p2 = fm(x[::-1], w1[::-1, -1])
    w2r, p2r = oracle_forward(om, x[::-1], w1[::-1, -1], DT, 10)
    assert np.abs(w2 - w2r).max() < 1e-6 and np.abs(p2 - p2r).max() < 1e-6
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[::-1][m], DT, 10, wsat0=w1[::-1, -1][m])
        assert np.abs(w2[m] - ref).max() <= 10 * noise + 1e-9
    with pytest.raises(ValueError):
        fm(x, w1[:2, -1])  # ragged ensembles: zip(strict=True) in utils.py:175


def test_single_member_sim_signature():
    """model.sim(dt, nTime, wsat0, pbar=False) -> (nTime+1, Nxy) (HistoryMatch.py:224)."""
    from oracle.ressim import set_perm

    om, gm = make_models(20, 20)
    x = perms(20, 20, 1, seed=13)[0]
    set_perm(om, x)
    gm.K = om.K
    ref = om.sim(DT, 12, np.zeros(400))
    out = gm.sim(DT, 12, np.zeros(400), pbar=False)
    assert out.shape == (13, 400) and out.dtype == np.float64
    _, noise = oracle_sim_and_noise(om, x, DT, 12)
    assert np.abs(out - ref).max() <= 10 * noise + 1e-9
    assert gm.actual_rates["inj"].shape == (1, 12) and gm.actual_rates["prd"].shape == (4, 12)


def test_unbalanced_rates_raise():
    _, gm = make_models(20, 20)
    gm.prd_rates = np.ones((4, 1)) / 3
    with pytest.raises(ValueError):
        gm.sim(DT, 2, np.zeros(400))


@pytest.mark.parametrize("variant", [1, 7, 9, 0])
def test_full_sim_128_within_reference_solver_noise(variant):
    """C2-shaped members (128x128): S after a few steps agrees with the oracle to within the spread the
    oracle itself shows when SuperLU's column ordering is changed (the reference's own numerical noise)."""
    n, N, steps = 128, 2, 3
    om, gm = make_models(n, n)
    x = perms(n, n, N, seed=17)
    plan = _plan(gm, N, nTime=steps)
    plan.set_variant(variant, variant)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    w, p, status = plan.outputs()
    assert not status.any()
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[m], DT, steps)
        err = np.abs(w[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (err, noise)
        assert err < 1e-4
    plan.close()


@pytest.mark.parametrize("nx,ny,variant", [(160, 160, 0), (96, 192, 0), (256, 256, 0), (256, 256, 15), (64, 256, 0)])
def test_large_grid_sim_cg_pressure(nx, ny, variant):
    """Ny > 128 (BASELINE configs 4/5 are 256x256 and 512x512).  256 x 256 (and 512 x 512) by default: the nested dissection of the
    larger grids, a direct solver like the reference's (HistoryMatch.py:362) -- no CG iterations are counted; other grids, and
    press_variant 15 at 256 x 256: the pressure system is solved by conjugate gradients (press_pcg.hip).
    Same acceptance as the 128x128 case: within the oracle's own solver noise."""
    N, steps = 2, 2
    om, gm = make_models(nx, ny)
    x = perms(nx, ny, N, seed=23)
    plan = _plan(gm, N, nTime=steps)
    plan.set_variant(variant, 0)
    plan.set_debug("embed", 0)  # (by default such a grid runs inside a 256 x 256 plan: test_grids_..._run_embedded...; here: the CG on the grid as given)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    assert not status.any()
    if (nx, ny, variant) == (256, 256, 0):
        assert st["mean_n_cg"] == 0  # direct
    else:
        assert 0 < st["mean_n_cg"] < 40 * max(nx, ny) + 1000
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[m], DT, steps)
        err = np.abs(w[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (err, noise)
        assert err < 1e-4
        assert np.array_equal(p[m], w[m][1:, om.xy2ind(*om.prd_xy.T)])  # obs_model gather, HistoryMatch.py:212-213
    plan.close()


def test_cg_solver_reports_non_convergence():
    n, N = 160, 2
    om, gm = make_models(n, n)
    plan = _plan(gm, N, nTime=1)
    plan.set_solver(rtol=1e-12, max_iter=5)
    plan.set_debug("embed", 0)  # the CG on the grid as given (embedded in a 256 x 256 plan the solve is direct)
    plan.set_inputs(perms(n, n, N, seed=5), transformed=False)
    plan.run()
    plan.sync()
    _, _, status = plan.outputs(want_wsats=False)
    assert (status & 8).all()  # HM_MEMBER_NO_CONVERGENCE
    plan.close()


@pytest.mark.parametrize("n,variant", [(20, 1), (128, 1), (128, 0), (256, 0)])
def test_general_fluid_parameters_bitexact(n, variant):
    """vw, vo, swc, sor away from the upstream defaults exercise the general RelPerm path (three extra divisions):
    assembly and saturation sweep stay bit-exact, the pressure solve stays within solver noise.  (256, 0): the slab sweep
    sat256s.hip in its general-fluid form against the oracle's sweep on the same fluxes."""
    from oracle.ressim import perm_transf, set_perm

    om, gm = make_models(n, n)
    for mdl in (om, gm):
        mdl.vw, mdl.vo, mdl.swc, mdl.sor = 0.7, 1.9, 0.05, 0.1
    N = 2
    x = perms(n, n, N, seed=23, scale=0.6)
    plan = _plan(gm, N, nTime=2)
    plan.set_variant(variant, variant)
    plan.set_inputs(perm_transf(x), transformed=True)
    S_in, Vxs, Vys, S_ref = [], [], [], []
    for m in range(N):
        set_perm(om, x[m])
        S = np.full(om.Nxy, om.swc)
        q, _, _ = om.source_field(0)
        _, Vx, Vy = om.pressure_step(S, q)
        S = om.saturation_step_upwind(S, q, Vx, Vy, DT)
        _, Vx, Vy = om.pressure_step(S, q)
        S_in.append(S), Vxs.append(Vx), Vys.append(Vy)
        S_ref.append(om.saturation_step_upwind(S, q, Vx, Vy, DT))
    plan.set_field("S", np.array(S_in))
    plan.pressure_only(0)
    Vx_gpu = plan.get_field("Vx")
    for m in range(N):
        assert np.abs(Vx_gpu[m] - Vxs[m]).max() < 1e-7
    plan.set_field("Vx", np.array(Vxs))
    plan.set_field("Vy", np.array(Vys))
    plan.saturation_only(0)
    S_gpu = plan.get_field("S").reshape(N, -1)
    for m in range(N):
        assert np.array_equal(S_gpu[m], S_ref[m])
    plan.close()


def test_time_varying_rates_and_single_member():
    """inj/prd rates with nTime columns (HistoryMatch.py:189-193, Optimise.py:760-767), N = 1."""
    from oracle.ressim import set_perm

    om, gm = make_models(20, 20)
    nT = 6
    rates = 0.5 + np.arange(nT) / nT
    for mdl in (om, gm):
        mdl.inj_rates = rates[None, :]
        mdl.prd_rates = np.tile(rates / 4, (4, 1))
    x = perms(20, 20, 1, seed=29)[0]
    set_perm(om, x)
    gm.K = om.K
    ref, noise = oracle_sim_and_noise(om, x, DT, nT)
    out = gm.sim(DT, nT, np.zeros(400), pbar=False)
    assert np.abs(out - ref).max() <= 10 * noise + 1e-9
    assert np.allclose(gm.actual_rates["inj"], rates[None, :])
    with pytest.raises(ValueError):
        gm.inj_rates = rates[None, :4]  # wrong number of columns
        gm.sim(DT, nT, np.zeros(400))


def test_nonuniform_porosity_uses_generic_path_and_matches():
    """Porosity field (upstream Gridded.por): per-cell pore volume enters the CFL count and dtx."""
    from oracle.ressim import set_perm

    om, gm = make_models(20, 20)
    por = 0.2 + 0.6 * np.random.RandomState(3).rand(20, 20)
    om.por = por
    gm.por = por
    x = perms(20, 20, 1, seed=31)[0]
    set_perm(om, x)
    gm.K = om.K
    ref = om.sim(DT, 8, np.zeros(400))
    out = gm.sim(DT, 8, np.zeros(400), pbar=False)
    assert np.abs(out - ref).max() < 1e-7
    assert abs(out[-1] @ (por.ravel() * om.h2) - 8 * DT) < 1e-9  # injected volume = water in place


@pytest.mark.parametrize("nx,ny,dtype,extras", [(20, 20, 64, ""), (20, 20, 32, ""), (16, 16, 64, "fluid"), (12, 20, 64, "porosity"), (25, 10, 64, ""),
                                                 (20, 20, 64, "chunks"), (8, 32, 64, "")])
def test_small_grid_single_launch_run_is_bit_identical(nx, ny, dtype, extras):
    """Small grids (the reference's default 20 x 20, HistoryMatch.py:97) run the whole forward pass as ONE launch, a wave per member
    (csrc/small.hip).  The kernel restates the generic kernels' arithmetic operation for operation (same sweeps, same order of the
    mat-vec partial sums), so a run through it equals the run through the generic kernels (variant 1 / 1) to the last bit: saturation
    history, producer series, sub-step counts, final pressures and fluxes; fp32 plans, a general fluid, a porosity field, non-square
    grids, a run continued in chunks from its own state."""
    N, steps = 5, 12
    _, gm = make_models(nx, ny, dtype=dtype)
    if extras == "fluid":
        gm.vw, gm.vo, gm.swc, gm.sor = 0.7, 1.9, 0.05, 0.1
    if extras == "porosity":
        gm.por = 0.2 + 0.3 * np.random.RandomState(3).rand(nx, ny)
    x = perms(nx, ny, N, seed=55)
    out = {}
    for variant in (1, 0):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(variant, variant)
        w0 = np.full((N, nx * ny), 0.05, dtype=np.float64 if dtype == 64 else np.float32) if extras == "fluid" else None
        plan.set_inputs(x, w0, transformed=False)
        if extras == "chunks" and variant == 0:
            plan.run(0, 5)
            plan.run(5, 1)
            plan.run(6, steps - 6)
        else:
            plan.run()
        st = plan.sync()
        w, p, status = plan.outputs()
        assert not status.any()
        out[variant] = (w, p, plan.get_field("nts"), plan.get_field("P"), plan.get_field("Vx"), plan.get_field("Vy"), plan.get_field("TX"))
        if variant == 0:
            assert st["ms_pressure"] == 0 and st["n_pressure_launches"] == steps  # one launch: no per-step kernels ran
        plan.close()
    assert out[0][0][:, -1].max() > 0.5
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b), np.abs(a.astype(float) - b.astype(float)).max()


def test_fp32_saturation_mode_tolerance():
    """dtype=32: saturation arithmetic and storage in fp32, pressure and Nts in fp64; <= 1e-3 abs on S."""
    from oracle.ressim import forward_model as oracle_forward

    om, gm = make_models(20, 20, dtype=32)
    x = perms(20, 20, 3, seed=37)
    from historymatching_amd.forward import make_forward_model

    w, p = make_forward_model(gm, DT, 20)(x)
    assert w.dtype == np.float32 and p.dtype == np.float32
    wr, pr = oracle_forward(om, x, None, DT, 20)
    assert np.abs(w - wr).max() < 1e-3 and np.abs(p - pr).max() < 1e-3


def _cell_xy(model, ix, iy):
    """Coordinates of the centre of cell (ix, iy)."""
    return [(ix + 0.5) * model.Lx / model.Nx, (iy + 0.5) * model.Ly / model.Ny]


@pytest.mark.parametrize("layout", ["default", "odd", "two_injectors", "injectors_share_a_band", "time_varying"])
def test_saturation_128_register_sweep_bitexact_vs_generic_and_image_sweeps(layout):
    """The three fp64 sweeps at 128 x 128 -- generic (sat_variant 1), fw image in LDS (sat128.hip, 5), fw in registers with scaled
    fluxes (sat128r.hip, the default) -- give identical saturations, producer series and sub-step counts over whole runs, for wells
    anywhere in a thread's 8 x 4 patch: injector and producers on patch corners, edges and interiors, on the first and last patch
    row (whose east / west terms cross the LDS halo), next to each other, two injectors in different bands; two injectors in
    one band of 16 rows, which the register sweep declines (the host falls back to the image sweep); rates that change per step."""
    n, N, steps = 128, 3, 5
    om, gm = make_models(n, n)
    if layout == "odd":  # injector in a patch interior (row 5, column 2), producers on patch row 7 / row 0 / a corner / beside the injector's patch
        cells = dict(inj=[(45, 70)], prd=[(7, 3), (120, 127), (15, 124), (47, 75)])
    elif layout == "two_injectors":
        cells = dict(inj=[(40, 17), (88, 100)], prd=[(3, 120), (125, 6), (64, 64)])
    elif layout == "injectors_share_a_band":
        cells = dict(inj=[(65, 20), (78, 100)], prd=[(3, 120), (125, 6)])
    else:
        cells = None
    if cells:
        gm.inj_xy = [_cell_xy(gm, *c) for c in cells["inj"]]
        gm.prd_xy = [_cell_xy(gm, *c) for c in cells["prd"]]
        gm.inj_rates = np.ones((len(cells["inj"]), 1)) / len(cells["inj"])
        gm.prd_rates = np.ones((len(cells["prd"]), 1)) / len(cells["prd"])
    if layout == "time_varying":
        r = 0.4 + np.arange(steps) / steps
        gm.inj_rates = r[None, :]
        gm.prd_rates = np.tile(r / 4, (4, 1))
    x = perms(n, n, N, seed=61)
    out = {}
    for sat_variant in (1, 5, 0):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(13, sat_variant)  # the same pressure kernel for all three
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, p, status = plan.outputs()
        assert not status.any()
        out[sat_variant] = (w, p, plan.get_field("nts"))
        plan.close()
    assert out[1][0][:, -1].max() > 0.5  # water did go in
    for v in (5, 0):
        for a, b in zip(out[1], out[v]):
            assert np.array_equal(a, b), (layout, v, np.abs(a - b).max())
        assert np.array_equal(np.signbit(out[1][0]), np.signbit(out[v][0]))


@pytest.mark.parametrize("general_fluid", [False, True])
def test_fp32_saturation_128_register_kernel_bitexact_vs_generic(general_fluid):
    """dtype=32 at 128x128: the register sweep (sat32s.hip: scaled fluxes, products through the clamp modifier, the compensated
    float32 pair of sat32.h) is bit-identical to k_saturation_generic<float> given the same fluxes (generic pressure kernel for
    both), and stays within the fp32 bar of the fp64 oracle."""
    n, N, steps = 128, 3, 3
    om, gm = make_models(n, n, dtype=32)
    if general_fluid:
        for mdl in (om, gm):
            mdl.vw, mdl.vo, mdl.swc, mdl.sor = 0.7, 1.9, 0.05, 0.1
    x = perms(n, n, N, seed=41)
    out = {}
    for sat_variant in (1, 0):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(1, sat_variant)
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, p, status = plan.outputs()
        assert not status.any() and w.dtype == np.float32
        out[sat_variant] = (w, p, plan.get_field("nts"))
        plan.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    from oracle.ressim import set_perm

    set_perm(om, x[0])
    ref = om.sim(DT, steps, np.zeros(n * n) + (om.swc if general_fluid else 0.0))
    if not general_fluid:
        assert np.abs(out[0][0][0] - ref).max() < 1e-3


@pytest.mark.parametrize("nx,ny,dtype", [(20, 20, 64), (128, 128, 64), (160, 96, 64), (192, 160, 64), (70, 300, 64), (128, 128, 32), (96, 192, 32)])
def test_streaming_saturation_kernel_bitexact_vs_generic(nx, ny, dtype):
    """k_saturation_stream (coefficients and neighbour fractional flows re-derived instead of stored, one pass and one
    barrier per sub-step) and k_saturation_tiled (the sweep used beyond 128 x 128: the same with fw shared through LDS
    tiles) are bit-identical to k_saturation_generic on the same fluxes."""
    N, steps = 2, 2
    _, gm = make_models(nx, ny, dtype=dtype)
    x = perms(nx, ny, N, seed=29)
    out = {}
    for sat_variant in (1, 2, 3):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(9 if ny > 128 else 1, sat_variant)  # same pressure kernel for all runs
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, p, status = plan.outputs()
        assert not status.any()
        out[sat_variant] = (w, p, plan.get_field("nts"))
        plan.close()
    for v in (2, 3):
        assert np.array_equal(out[1][2], out[v][2]) and out[1][2].min() >= 1
        assert np.array_equal(out[1][0], out[v][0]) and np.array_equal(out[1][1], out[v][1])


@pytest.mark.parametrize("nx,ny,N,wells,dtype", [(256, 256, 3, "default", 64), (256, 128, 2, "default", 64), (128, 256, 2, "default", 64),
                                                 (384, 256, 2, "edges", 64), (256, 256, 67, "default", 64),
                                                 (256, 256, 3, "default", 32), (384, 256, 2, "edges", 32), (128, 256, 67, "default", 32)])
def test_multi_tile_saturation_teams_bitexact_vs_tiled(nx, ny, N, wells, dtype):
    """Grids made of 128 x 128 tiles run the saturation sweep as teams of workgroups (sat128t.hip: one workgroup per tile, tile edges
    exchanged once per sub-step; fp64 grids 256 cells wide by default as slabs of 64 rows with the fractional flow in registers,
    sat256s.hip: first and last wave of a slab trade one row per sub-step; dtype = 32 plans: slabs of 16 384 cells, sat32s.hip).  Saturations, producer series and sub-step counts are bit-identical to the
    single-workgroup tiled kernel (sat_variant 3).  The default injector sits on a tile corner (its exact fractional flow
    travels in the published edges); "edges" puts wells on every kind of tile border; 67 members = more members than
    teams that fit the chip at once (a second, partial round)."""
    steps = 1 if N > 8 else 2
    _, gm = make_models(nx, ny, dtype=dtype)
    if wells == "edges":
        hx, hy = gm.Lx / nx, gm.Ly / ny
        cells = [(127, 40), (128, 200), (200, 127), (300, 128), (255, 255), (256, 0)]   # last/first rows and columns of tiles
        gm.inj_xy = [[(ix + 0.5) * hx, (iy + 0.5) * hy] for ix, iy in cells[:2]]
        gm.prd_xy = [[(ix + 0.5) * hx, (iy + 0.5) * hy] for ix, iy in cells[2:]]
        gm.inj_rates = np.ones((2, 1)) / 2
        gm.prd_rates = np.ones((4, 1)) / 4
    x = perms(nx, ny, N, seed=37)
    out = {}
    for sat_variant in (0, 5, 3):  # 0: the default (fp64 grids 256 wide: slabs, sat256s.hip; else tile teams), 5: tile teams, 3: tiled
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(0, sat_variant)
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, p, status = plan.outputs()
        assert not status.any()
        out[sat_variant] = (w, p, plan.get_field("nts"))
        plan.close()
    for v in (0, 5):
        assert np.array_equal(out[v][2], out[3][2]) and out[v][2].min() >= 1
        assert np.array_equal(out[v][0], out[3][0]) and np.array_equal(out[v][1], out[3][1])
    assert out[0][0][:, -1].max() > 0.5  # the front has left the injector


def test_float32_slab_sweep_skips_dry_slabs_without_changing_a_bit():
    """The float32 slab sweep launches the whole ensemble at once and lets the workgroups of slabs that are dry (with dry neighbours and no
    injector) leave at once (sat32s.hip: ACTIVE slabs).  512 x 256 (Nx = 512: 16 slabs of 32 rows... here 8 slabs of 64 rows at Ny = 256), 20
    members, 10 steps: saturation history, producer series and sub-step counts equal those of round 4's form -- every slab of every member
    in rounds of co-resident teams, hm_fwd_set_debug "team_rounds" -- and of the single-workgroup tiled kernel; a run continued from a
    state written by hand (the record of wet slabs is void: every slab takes part) as well."""
    nx, ny, N, steps = 512, 256, 20, 10
    _, gm = make_models(nx, ny, dtype=32)
    x = perms(nx, ny, N, seed=91)
    out = {}
    for mode in ("single", "rounds", "tiled"):
        plan = _plan(gm, N, nTime=steps + 2)
        if mode == "rounds":
            plan.set_debug("team_rounds", 1)
        if mode == "tiled":
            plan.set_variant(0, 3)
        plan.set_inputs(x, transformed=False)
        plan.run(0, steps)
        S = plan.get_field("S")
        S[:, 300:310, 17:40] = 0.3   # water far from the front, in slabs that sat the last steps out
        plan.set_field("S", S)
        plan.run(steps, 2)
        st = plan.sync()
        w, p, status = plan.outputs()
        assert not status.any() and st["team_retries"] == 0  # nobody waited in vain, nothing went to the tiled sweep
        out[mode] = (w, p, plan.get_field("nts"), st["slab_redos"])
        plan.close()
    assert out["single"][0][:, steps].max() > 0.5 and out["tiled"][3] == 0
    dry_rows = (out["single"][0][:, steps].reshape(N, nx, ny) == 0).all(axis=(0, 2))
    assert dry_rows[:64].all() and dry_rows[-64:].all()  # the outer slabs are indeed still dry after ten steps: they did sit out
    for mode in ("rounds", "tiled"):
        for a, b in zip(out["single"][:3], out[mode][:3]):
            assert np.array_equal(a, b), (mode, np.abs(a.astype(float) - b.astype(float)).max())


@pytest.mark.parametrize("n", [20, 128, 256])
def test_anisotropic_permeability(n):
    """K = (Kx, Ky) with Kx != Ky (the simulator's K is (2, Nx, Ny), HistoryMatch.py:164; the reference itself stacks Kx = Ky):
    TX from Kx, TY from Ky, the SPD pin Kx[0,0] + Ky[0,0].  `model.sim` with an anisotropic K against the oracle
    within its solver noise (generic kernels at 20 x 20, matrix-core solver at 128 x 128, two-level CG at 256 x 256), assembly
    bit-exact, and the ensemble form (`ForwardPlan.set_inputs(..., perms_y=)`) equal to the single-member runs."""
    import oracle.ressim as orc
    from scipy.sparse.linalg import spsolve

    steps = 3 if n <= 128 else 2
    om, gm = make_models(n, n)
    x = perms(n, n, 4, seed=71)
    Kx = 0.1 + np.exp(5 * x[:2])
    Ky = 0.1 + np.exp(5 * (0.6 * x[2:] - 0.3))
    outs = []
    for m in range(2):
        K = np.stack([Kx[m].reshape(n, n), Ky[m].reshape(n, n)])
        om.K = K
        gm.K = K
        ref = om.sim(DT, steps, np.zeros(n * n))
        orig = orc.spsolve
        orc.spsolve = lambda A, b: spsolve(A.tocsc(), b, permc_spec="NATURAL" if n <= 128 else "MMD_AT_PLUS_A")
        try:
            noise = float(np.abs(om.sim(DT, steps, np.zeros(n * n)) - ref).max())
        finally:
            orc.spsolve = orig
        out = gm.sim(DT, steps, np.zeros(n * n))
        err = np.abs(out - ref).max()
        assert err <= 10 * noise + 1e-9, (m, err, noise)
        outs.append(out)
    # different from the isotropic run with Kx alone (the y-permeability matters)
    gm.K = Kx[0].reshape(n, n)
    assert np.abs(gm.sim(DT, steps, np.zeros(n * n)) - outs[0]).max() > 1e-3
    plan = _plan(gm, 2, nTime=steps)
    plan.set_inputs(Kx, None, transformed=True, perms_y=Ky)
    plan.pressure_only(0)
    TX, TY = plan.get_field("TX"), plan.get_field("TY")
    hx, hy = gm.Lx / n, gm.Ly / n
    for m in range(2):  # S = 0: Mt = 1, L = 1/K
        Lx, Ly = 1.0 / Kx[m].reshape(n, n), 1.0 / Ky[m].reshape(n, n)
        assert np.array_equal(TX[m][1:-1, :], 2 * hy / hx / (Lx[:-1, :] + Lx[1:, :]))
        assert np.array_equal(TY[m][:, 1:-1], 2 * hx / hy / (Ly[:, :-1] + Ly[:, 1:]))
    plan.run()
    plan.sync()
    w, _, status = plan.outputs()
    plan.close()
    assert not status.any()
    for m in range(2):
        if n <= 128:
            assert np.array_equal(w[m], outs[m])
        else:  # the CG passes of a member are split over a member-count-dependent number of workgroups: rounding-level differences
            assert np.abs(w[m] - outs[m]).max() < 1e-6


@pytest.mark.parametrize("dtype", [64, 32])
def test_tile_team_timeout_is_retried_by_the_tiled_sweep(dtype):
    """A tile team that gives up waiting for a neighbour (its workgroups were not all resident: CUs held by another process)
    flags the member HM_MEMBER_SYNC_TIMEOUT; the host redoes that time step with the single-workgroup tiled sweep instead of
    failing the member.  saturation variant 4 takes the retry path on every step: same result as the teams, status clean."""
    n, N, steps = 256, 3, 2
    _, gm = make_models(n, n, dtype=dtype)
    x = perms(n, n, N, seed=37)
    out = {}
    for sat_variant in (0, 4):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(0, sat_variant)
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, p, status = plan.outputs()
        assert not status.any()
        out[sat_variant] = (w, p, plan.get_field("nts"))
        plan.close()
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[4]))


@pytest.mark.parametrize("nx,ny,dtype", [(100, 100, 64), (64, 96, 64), (40, 24, 64), (48, 128, 64), (100, 100, 32), (33, 57, 32),
                                         (160, 160, 64), (96, 192, 64), (200, 150, 32), (300, 260, 64), (260, 300, 32)])
def test_grids_run_embedded_in_the_next_square_the_fast_kernels_take(nx, ny, dtype):
    """A grid that no specialised kernel takes (not 128 / 256 / 512 square, not made of 128-wide blocks, not small enough for the
    one-launch kernel) runs inside a 128 x 128, 256 x 256 or 512 x 512 plan,
    padded with cells of zero permeability (csrc/forward.hip: embedded grids).  Against the generic kernels on the grid as given
    (hm_fwd_set_debug "embed" 0) and against the oracle:
      * the sweep for given fluxes: the same bits as the oracle (fp64) / as the generic kernels (fp32), the same sub-step counts;
      * transmissibilities out of the embedded plan: the same bits; pressure and fluxes: within the solvers' noise of the generic pair;
      * a run of several steps: within the oracle's own solver noise (fp64), producer series = the history at the producers."""
    from oracle.ressim import perm_transf, set_perm

    N, steps = (3, 3) if max(nx, ny) <= 128 else (2, 2) if max(nx, ny) <= 256 else (2, 1)
    om, gm = make_models(nx, ny, dtype=dtype)
    x = perms(nx, ny, N, seed=11)
    plans = {}
    for name in ("embedded", "generic"):
        plan = _plan(gm, N, nTime=steps)
        if name == "generic":
            plan.set_debug("embed", 0)
        plan.set_inputs(x, transformed=False)
        plans[name] = plan
    # --- one sweep from given fluxes (the oracle's, from a developed state)
    S_in, Vxs, Vys, S_ref, nts_ref = [], [], [], [], []
    for m in range(N):
        S, q = _oracle_state(om, x[m], 2 if max(nx, ny) <= 128 else 1)
        set_perm(om, x[m])
        _, Vx, Vy = om.pressure_step(S, q)
        S_in.append(S), Vxs.append(Vx), Vys.append(Vy)
        S_ref.append(om.saturation_step_upwind(S, q, Vx, Vy, DT))
        nts_ref.append(om.cfl_substeps(Vx, Vy, q, DT)[0])
    ft = np.float64 if dtype == 64 else np.float32
    out = {}
    for name, plan in plans.items():
        plan.set_field("S", np.array(S_in).astype(ft))
        plan.set_field("Vx", np.array(Vxs))
        plan.set_field("Vy", np.array(Vys))
        plan.saturation_only(0)
        out[name] = (plan.get_field("S").reshape(N, -1), plan.get_field("nts")[:, 0].copy())
        assert out[name][1].tolist() == nts_ref
    assert np.array_equal(out["embedded"][0], out["generic"][0])
    if dtype == 64:
        for m in range(N):
            assert np.array_equal(out["embedded"][0][m], S_ref[m])
    # --- one pressure step from that state
    fields = {}
    for name, plan in plans.items():
        plan.set_field("S", np.array(S_in).astype(ft))
        plan.pressure_only(0)
        fields[name] = {k: plan.get_field(k) for k in ("TX", "TY", "P", "Vx", "Vy")}
    assert np.array_equal(fields["embedded"]["TX"], fields["generic"]["TX"]) and np.array_equal(fields["embedded"]["TY"], fields["generic"]["TY"])
    tol = 1e-8 if max(nx, ny) <= 128 else 1e-6  # (above 128 the generic pair solves by conjugate gradients to rtol 1e-12 of the residual)
    for k in ("Vx", "Vy"):
        assert np.abs(fields["embedded"][k] - fields["generic"][k]).max() < tol, k
    pe, pg = fields["embedded"]["P"].reshape(N, -1), fields["generic"]["P"].reshape(N, -1)
    assert np.abs(pe - pg).max() < 10 * tol * np.abs(pg).max()
    # --- a run
    res = {}
    for name, plan in plans.items():
        plan.set_inputs(x, transformed=False)
        plan.run()
        plan.sync()
        w, pr, status = plan.outputs()
        assert not status.any()
        res[name] = (w, pr, plan.get_field("nts").copy())
    assert np.array_equal(res["embedded"][2], res["generic"][2])
    for m in range(N):
        if dtype == 64:
            ref, noise = oracle_sim_and_noise(om, x[m], DT, steps)
            assert np.abs(res["embedded"][0][m] - ref).max() <= 10 * noise + 1e-9
        else:
            assert np.abs(res["embedded"][0][m].astype(np.float64) - res["generic"][0][m]).max() < 1e-4
    assert np.abs(res["embedded"][1].astype(np.float64) - res["generic"][1]).max() < (1e-7 if dtype == 64 else 1e-4)
    for plan in plans.values():
        plan.close()


@pytest.mark.parametrize("nx,ny", [(64, 128), (200, 128), (128, 64)])
def test_rectangular_grids_mixed_kernels(nx, ny):
    """Ny = 128 with Nx != 128 runs the matrix-core pressure solver (any number of 128-wide blocks) with the generic
    saturation sweep; Ny != 128 runs the generic pair.  Same acceptance as the square case."""
    N, steps = 2, 2
    om, gm = make_models(nx, ny)
    x = perms(nx, ny, N, seed=31)
    plan = _plan(gm, N, nTime=steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    w, p, status = plan.outputs()
    assert not status.any()
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[m], DT, steps)
        err = np.abs(w[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (err, noise)
    plan.close()


def test_whole_run_40_steps_at_config2_shape():
    """BASELINE config 2 shape (128x128, nTime = 40, dt = 0.025, wells of HistoryMatch.py:177-190): members of the synthetic
    prior through the default kernels for the full run.  The per-step CFL sub-step counts of GPU and oracle are EQUAL (the
    `ceil` of SURVEY.md A.4 never flips: its argument is 614.4, set by the injector cell), so the comparison is between two
    direct solvers of the same systems and the bar is SURVEY.md 8d's 1e-8 -- for a member whose systems are well conditioned.
    Member 2 of this prior is not: K spans 0.1 .. 1.2e6, cond_1(A) ~ 4e12, SuperLU's own two column orderings differ by
    4e-6 in pressure and 1.5e-7 in flux after ONE solve (residuals 4e-9 and 1.6e-8) and by 1.1e-5 in saturation after 40
    steps with identical sub-step counts (profiles/r03/long_parity.txt); no fp64 solver can be asked for more than that
    spread there, so for such a member -- named here, not hidden in a wide bar -- the bar is 10 x the oracle's own spread."""
    from tests.helpers import oracle_sim_noise_and_nts

    n, steps = 128, 40
    om, gm = make_models(n, n)
    x = perms(n, n, 3, seed=1)[[0, 2]]
    plan = _plan(gm, 2, nTime=steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and st["mean_nts"] > 100
    ill_conditioned = {1}  # row 1 = member 2 of the seed-1 prior
    for m in range(2):
        ref, noise, nts_o = oracle_sim_noise_and_nts(om, x[m], DT, steps)
        assert np.array_equal(nts[m], nts_o), f"member {m}: sub-step counts differ at steps {np.flatnonzero(nts[m] != nts_o)}"
        err = np.abs(w[m] - ref).max()
        if m in ill_conditioned:
            assert noise > 1e-7, "member 2 is the ill-conditioned case this test names; if it no longer is, tighten its bar"
            assert err <= 10 * noise, (m, err, noise)
        else:
            assert noise < 1e-9 and err <= 1e-8, (m, err, noise)
        assert np.array_equal(p[m], w[m][1:, om.xy2ind(*om.prd_xy.T)])


def test_nested_dissection_pressure_whole_run():
    """press_variant 12 (press_nd.hip: nested-dissection factorisation of the same TPFA system; the default at
    128 x 128) through a 20-step run: within the oracle's own solver noise at every stored step, same sub-step counts as the block
    elimination (variant 13), and a second ensemble on the same plan reproduces the first bit for bit (no state leaks between runs)."""
    n, steps, N = 128, 20, 3
    om, gm = make_models(n, n)
    x = perms(n, n, N, seed=5)
    res = {}
    for v in (13, 12):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(v, 0)
        plan.set_inputs(x, transformed=False)
        plan.run()
        st = plan.sync()
        w, p, status = plan.outputs()
        res[v] = (w.copy(), st["mean_nts"])
        assert not status.any()
        if v == 12:
            plan.set_inputs(x, transformed=False)
            plan.run()
            plan.sync()
            w2, _, _ = plan.outputs()
            assert np.array_equal(w, w2)
        plan.close()
    assert res[13][1] == res[12][1], "the two direct solvers must give the same CFL sub-step counts"
    for m in range(N):
        ref, noise = oracle_sim_and_noise(om, x[m], DT, steps)
        assert np.abs(res[12][0][m] - ref).max() <= 10 * noise + 1e-9, (m, np.abs(res[12][0][m] - ref).max(), noise)


@pytest.mark.parametrize("n,rates", [(128, "constant"), (128, "time_varying"), (128, "piecewise_constant"), (256, "constant"), (256, "piecewise_constant"),
                                     (512, "constant")])
def test_nested_dissection_reuse_of_dry_fronts_is_bit_identical(n, rates):
    """press_nd.hip does not eliminate a front again while its whole subtree is still dry and the rates of the wells in it are unchanged
    (k_nd_plan): the stored factor rows and update matrix are what it would recompute.  Whole runs with the reuse (press_variant 12, the default) and without (14) give
    array_equal saturations, producer series, sub-step counts and final pressures; so does a second run on the SAME plan from other
    permeabilities (the cache dies with the inputs it was computed from) and a pressure solve after a saturation field was written
    into the plan by hand.  256 x 256 / 512 x 512: the same plan for the 13- / 15-level tree (k_ndl_plan: lists for the wave-level fronts,
    `todo` bytes for the workgroup-level and big fronts)."""
    N, steps = (4, 14) if n < 512 else (3, 8)
    _, gm = make_models(n, n)
    if rates != "constant":
        # the right-hand side rows of a front with a well in its subtree are kept only over runs of time steps with equal rates
        r = 0.5 + np.arange(steps) / steps if rates == "time_varying" else np.where(np.arange(steps) < 5, 0.7, np.where(np.arange(steps) < 9, 1.3, 0.9))
        gm.inj_rates = r[None, :]
        gm.prd_rates = np.tile(r / 4, (4, 1))
    x = perms(n, n, 2 * N, seed=71)
    out = {}
    for v in (14, 12):
        plan = _plan(gm, N, nTime=steps)
        plan.set_variant(v, 0)
        runs = []
        for xs in (x[:N], x[N:]):  # two runs on one plan
            plan.set_inputs(xs, transformed=False)
            plan.run()
            plan.sync()
            w, p, status = plan.outputs()
            assert not status.any()
            runs.append((w.copy(), p.copy(), plan.get_field("nts").copy(), plan.get_field("P").copy()))
        # a saturation field written by hand: wet in one corner only, then one pressure solve, twice (the second from the cache)
        S = np.zeros((N, n, n))
        S[:, :20, :30] = 0.4
        plan.set_field("S", S.reshape(N, -1))
        plan.pressure_only(0)
        P1 = plan.get_field("P").copy()
        plan.pressure_only(0)
        runs.append((P1, plan.get_field("P").copy()))
        out[v] = runs
        plan.close()
    assert out[12][0][0][:, -1].max() > 0.5  # water did go in
    for a_run, b_run in zip(out[14], out[12]):
        for a, b in zip(a_run, b_run):
            assert np.array_equal(a, b), np.abs(a - b).max()
    assert np.array_equal(out[12][2][0], out[12][2][1])  # the solve from the cache equals the one that filled it


def test_nested_dissection_reuse_ends_when_a_raw_field_pointer_is_handed_out():
    """A device pointer to K taken with hm_fwd_device_ptr can be written at any later time without the library seeing it
    (device-to-device chaining): from that call on the plan keeps no factor rows across solves (fwd.h raw_field_exposed), so a solve
    after such a write equals the same solve on a fresh plan."""
    import ctypes as C

    from historymatching_amd import _lib

    n, N = 128, 3
    _, gm = make_models(n, n)
    x = perms(n, n, 2 * N, seed=73)
    S = np.zeros((N, n, n))
    S[:, :20, :30] = 0.4
    ref = _plan(gm, N, nTime=2)
    ref.set_inputs(x[N:], transformed=False)
    ref.set_field("S", S.reshape(N, -1))
    ref.pressure_only(0)
    P_ref = ref.get_field("P").copy()
    K2 = np.ascontiguousarray(ref.get_field("K"))
    ref.close()

    plan = _plan(gm, N, nTime=2)
    plan.set_inputs(x[:N], transformed=False)
    plan.set_field("S", S.reshape(N, -1))
    plan.pressure_only(0)
    ptr = plan.device_ptr("K")
    plan.pressure_only(0)  # would refill the cache under the generation the call above started
    P1 = plan.get_field("P").copy()
    _lib.check(plan.lib.hm_copy_to_device(plan.ctx.handle, C.c_void_p(ptr), K2.ctypes.data_as(C.c_void_p), K2.nbytes), "copy K")
    plan.pressure_only(0)
    P2 = plan.get_field("P").copy()
    plan.close()
    assert not np.array_equal(P1, P2)
    assert np.array_equal(P2, P_ref), np.abs(P2 - P_ref).max()


def test_comp1_single_member_composite():
    """`comp1(perm, wsat0)` (HistoryMatch.py:358-364) = a batch of one through the same device path: identical to the
    member's row of the ensemble run."""
    from historymatching_amd.forward import make_forward_model

    _, gm = make_models(20, 20)
    x = perms(20, 20, 3, seed=2)
    fm = make_forward_model(gm, DT, 6)
    w, p = fm(x)
    w1, p1 = fm.comp1(x[1])
    assert w1.shape == (7, 400) and p1.shape == (6, 4)
    assert np.array_equal(w1, w[1]) and np.array_equal(p1, p[1])


def test_fp32_mode_default_kernels_128():
    """dtype=32 at 128x128 through the DEFAULT kernels (matrix-core pressure solver reading the fp32 saturation, fp32
    register-resident sweep): within the fp32 bar of the fp64 oracle (<= 1e-3 abs on S), same Nts."""
    n, N, steps = 128, 2, 4
    om, gm = make_models(n, n, dtype=32)
    x = perms(n, n, N, seed=43)
    plan = _plan(gm, N, nTime=steps)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and w.dtype == np.float32
    from oracle.ressim import set_perm

    for m in range(N):
        set_perm(om, x[m])
        ref = om.sim(DT, steps, np.zeros(n * n))
        assert np.abs(w[m] - ref).max() < 1e-3
        assert np.abs(p[m] - ref[1:, om.xy2ind(*om.prd_xy.T)]).max() < 1e-3
    assert nts.min() >= 100


def test_fp32_mode_128_whole_run_vs_fp64_oracle():
    """dtype=32 plans over a WHOLE run at config 3's grid: four members, 40 steps (24 600 explicit sub-steps), default kernels, against
    the fp64 oracle: <= 1e-3 on every saturation and producer value of the history, same sub-step counts.  (A plain
    float32 accumulator drifts to 5e-4 here and to 2.4e-2 / 0.30 on the larger grids: profiles/r05/fp32_drift_*_before.txt; the
    compensated pair of csrc/sat32.h: 6e-5.)"""
    from tests.helpers import oracle_sims_and_noise_parallel

    n, N = 128, 4
    om, gm = make_models(n, n, dtype=32)
    x = perms(n, n, N, seed=43)
    plan = _plan(gm, N, nTime=NT)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    w, p, status = plan.outputs()
    nts = plan.get_field("nts")
    plan.close()
    assert not status.any() and w.dtype == np.float32 and (nts == 615).all()
    worst = 0.0
    for m, (ref, noise) in enumerate(oracle_sims_and_noise_parallel(n, n, x, DT, NT)):
        err = np.abs(w[m] - ref).max(axis=1)
        worst = max(worst, float(err.max()))
        assert err.max() < 1e-3, (m, err.max(), noise)
        assert np.abs(p[m] - ref[1:, om.xy2ind(*om.prd_xy.T)]).max() < 1e-3
        assert abs(float(w[m][-1].astype(float).mean()) - float(ref[-1].mean())) < 1e-6  # water in place
    assert worst < 2e-4  # observed 4e-5 (the front amplifies single-precision rounding of the fractional flow)


def test_config2_full_size_properties():
    """BASELINE config 2 at full size (N_e = 1000, 128 x 128, 40 steps; too large for the oracle in a test) through
    size-independent properties of the scheme:
    (1) mass balance: the explicit upwind sweep is conservative, so with unit porosity the water in place equals the injected
        volume minus the produced water.  Production of a step is dt q_prd fw(S_prd) with S_prd between its values at the two
        ends of the step (producer saturations only grow), which brackets the water in place from the stored producer series;
    (2) 0 <= S <= 1;
    (3) members are independent: a sub-ensemble run alone reproduces its members bit for bit;
    (4) the producer series is the saturation history at the producer cells."""
    n, N, steps = 128, 1000, 40
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=1)
    plan = _plan(gm, N, nTime=steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    plan.run()
    plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    assert not status.any()
    assert S_end.shape == (N, n * n) and prods.shape == (N, steps, 4)
    assert S_end.min() >= 0.0 and S_end.max() <= 1.0 and np.isfinite(prods).all()
    assert (np.diff(prods, axis=1) >= -1e-12).all()                     # producer saturations only grow
    fw = lambda s: s * s / (s * s + (1 - s) * (1 - s))                  # noqa: E731  (vw = vo = 1, swc = sor = 0)
    water = S_end.sum(1) * (gm.Lx / n) * (gm.Ly / n)
    injected = steps * DT * 1.0
    at_end = DT * 0.25 * fw(prods).sum((1, 2))                          # >= the water produced
    at_start = DT * 0.25 * fw(np.concatenate([np.zeros((N, 1, 4)), prods[:, :-1]], 1)).sum((1, 2))   # <= the water produced
    assert (water >= injected - at_end - 1e-9).all() and (water <= injected - at_start + 1e-9).all()
    assert (at_end > at_start).mean() > 0.5                             # most members have seen breakthrough: the bracket is not trivial
    # (3) + (4) on a sub-ensemble with history
    sub = [0, 17, 503, 999]
    plan = _plan(gm, len(sub), nTime=steps)
    plan.set_inputs(x[sub], transformed=False)
    plan.run()
    plan.sync()
    w, p, st2 = plan.outputs()
    plan.close()
    assert not st2.any()
    assert np.array_equal(w[:, -1, :], S_end[sub]) and np.array_equal(p, prods[sub])
    assert np.array_equal(p, w[:, 1:, gm.xy2ind(*gm.prd_xy.T)])


@pytest.mark.parametrize("dtype", [64, 32])
def test_large_grid_whole_run_properties(dtype):
    """256 x 256 (config 4's grid), 70 members (two rounds of workgroup teams), all 40 steps, default kernels (nested-dissection
    pressure solve + slab / tile-team sweep): the mass-balance bracket of test_config2_full_size_properties, bounds, and member
    independence across team rounds: a sub-ensemble run alone gives the same members bit for bit (the direct solver treats every
    member by itself, in a fixed order of operations)."""
    n, N, steps = 256, 70, 40
    _, gm = make_models(n, n, dtype=dtype)
    x = perms(n, n, N, seed=2)
    plan = _plan(gm, N, nTime=steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    assert not status.any() and st["mean_n_cg"] == 0
    S_end = S_end.astype(float)
    tol = 1e-9 if dtype == 64 else 2e-4
    assert S_end.min() >= -tol and S_end.max() <= 1.0 + tol
    fw = lambda s: s * s / (s * s + (1 - s) * (1 - s))  # noqa: E731
    water = S_end.sum(1) * (gm.Lx / n) * (gm.Ly / n)
    injected = steps * DT * 1.0
    p64 = prods.astype(float)
    at_end = DT * 0.25 * fw(p64).sum((1, 2))
    at_start = DT * 0.25 * fw(np.concatenate([np.zeros((N, 1, 4)), p64[:, :-1]], 1)).sum((1, 2))
    assert (water >= injected - at_end - tol).all() and (water <= injected - at_start + tol).all()
    sub = [3, 64, 69]  # members of the first and of the second round
    plan = _plan(gm, len(sub), nTime=steps, keep_history=False)
    plan.set_inputs(x[sub], transformed=False)
    plan.run()
    plan.sync()
    S_sub, p_sub, st2 = plan.outputs()
    plan.close()
    assert not st2.any()
    assert np.array_equal(S_sub, S_end[sub].astype(S_sub.dtype)) and np.array_equal(p_sub, prods[sub])


def test_direct_solver_survives_an_ill_conditioned_member():
    """256 x 256, nested dissection (press_nd256.o).  An elimination without pivoting cannot solve a member whose permeability spans
    more than ten orders of magnitude (cond(A) beyond 1 / eps: the diagonal of a strongly coupled cluster cancels to nothing) -- found
    on one of the 4096 members of BASELINE config 4 (K = 0.1 ... 1.2e9).  The library checks every solve a posteriori (a non-positive
    pivot, or fluxes that miss the wells by more than 1e-4 of the largest rate: k_nd_flux) and solves such a member again for that time
    step with the two-level CG (nd_check_and_fall_back); the reference's sparse direct solver with partial pivoting does not fail
    there either (HistoryMatch.py:362).  The run must complete with clean status words and finite results; the healthy member beside it
    must not notice (bit-identical to a run without the pathological one)."""
    from historymatching_amd.geostat import gaussian_fields_kron

    n, steps = 256, 40
    _, gm = make_models(n, n)
    x = np.concatenate([_config4_member_2086(n), perms(n, n, 1, seed=5).astype(np.float32).astype(np.float64)])
    assert 1.1e9 < (0.1 + np.exp(5 * x[0])).max() < 1.3e9
    plan = _plan(gm, 2, nTime=steps, keep_history=False)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    S_end, prods, status = plan.outputs()
    plan.close()
    assert not status.any(), status
    # (whether and when this member trips the check depends on the last bits of the solve: 8 hand-overs from step 32 on with one order of
    # the back substitution's sums, none within 40 steps with another -- the hand-over itself is exercised deterministically below)
    # What is asked of such a member is what the reference's solver delivers on it: the run completes with finite numbers and no flag.
    # Its saturations are NOT asserted to stay in [0, 1]: with fluxes of 1e9 (p_c - p_nb) the mass balance of ANY fp64 pressure field is
    # only good to ~1e-5 per cell and step (820 of saturation per unit of flux error at this cell size), for the CG as for a direct solver.
    assert np.isfinite(S_end).all() and np.isfinite(prods).all()
    assert S_end[1].min() >= -1e-9 and S_end[1].max() <= 1 + 1e-9
    plan = _plan(gm, 1, nTime=steps, keep_history=False)
    plan.set_inputs(x[1:], transformed=False)
    plan.run()
    st1 = plan.sync()
    S1, _, status1 = plan.outputs()
    plan.close()
    assert not status1.any() and st1["nd_fallbacks"] == 0
    assert np.array_equal(S1[0], S_end[1])


def test_large_grid_run_is_asynchronous_without_the_a_posteriori_check():
    """press_variant 12 at 256 x 256 (the direct solver without its per-step check) queues the whole run and returns: nothing in
    hm_fwd_run waits for the device -- the team sweeps' time-out retry is a device-gated launch since round 5 (forward.hip:
    launch_saturation), the solver's buffers are set up before the run.  The call returns in a small fraction of the time the device then
    needs; the default variant (0) reads the status words back once per time step (the hand-over to the two-level CG is a host decision:
    a device-side single-workgroup CG would cost 0.4-0.5 s per member-step, profiles/r05/pcg_on_hard_member.txt) and so takes as long as
    the run.  Same results from both."""
    import time

    n, N, steps = 256, 64, 12
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=19)
    out = {}
    for variant in (12, 0):
        plan = _plan(gm, N, nTime=steps, keep_history=False)
        plan.set_variant(variant, 0)
        plan.set_inputs(x, transformed=False)
        plan.run(0, 1)  # first step: buffers, tables
        plan.sync()
        t0 = time.perf_counter()
        plan.run(1, steps - 1)
        t_call = time.perf_counter() - t0
        plan.sync()
        t_all = time.perf_counter() - t0
        S, p, status = plan.outputs()
        plan.close()
        assert not status.any()
        out[variant] = (S, p, t_call, t_all)
    assert out[12][2] < 0.25 * out[12][3], (out[12][2], out[12][3])   # queued, not waited for
    assert out[0][2] > 0.8 * out[0][3]                                  # the default: one read-back per time step
    assert np.array_equal(out[12][0], out[0][0]) and np.array_equal(out[12][1], out[0][1])


def _config4_member_2086(n=256):
    """Row 2086 of gaussian_fields_kron(256, 256, 2, 1, 4096, r=0.8, seed=1000) (bench.py: config4_sharded), rounded to fp32 as the device-resident
    assimilation hands it over: K = 0.1 ... 1.2e9.  The normals of the rows before it are drawn and dropped."""
    from historymatching_amd.geostat import gaussian_fields_kron

    rng = np.random.RandomState(1000)
    for _ in range(2086 // 149):  # 2086 = 14 * 149
        rng.randn(149, n, n)
    return gaussian_fields_kron(n, n, 2, 1, 1, r=0.8, rng=rng).astype(np.float32).astype(np.float64)


def test_ill_conditioned_member_pressure_solves_vs_the_oracles_spsolve():
    """The one member of config 4's prior on which the elimination without pivoting eventually breaks down (K = 0.1 ... 1.2e9), measured
    against the reference's kind of solver where it matters: the face fluxes of the nested dissection WITHOUT its a-posteriori check
    (press_variant 12) and of the two-level CG (15) against oracle.ressim's spsolve on the same saturation states -- the state after 0, 10
    and 25 steps of the default run -- with the oracle's own spread between two SuperLU orderings as the yardstick, as for the healthy
    members (test_assembly_bitexact_and_pressure_within_solver_noise).  While the check of the default path does not trip, the direct
    solve is as close to the oracle as the oracle is to itself (bar: 10 x that spread); the divergence residual both solvers leave is
    the T eps |p| of any fp64 solver on this K."""
    from oracle.ressim import perm_transf, set_perm
    from scipy.sparse.linalg import spsolve

    n = 256
    om, gm = make_models(n, n)
    x = np.concatenate([_config4_member_2086(n), perms(n, n, 1, seed=5)])
    set_perm(om, x[0])
    q = om.source_field(0)[0]
    plan = _plan(gm, 2, nTime=26, keep_history=True)
    plan.set_inputs(x, transformed=False)
    plan.run()
    st = plan.sync()
    w, _, status = plan.outputs()
    plan.close()
    assert not status.any()
    rows = []
    for k in (0, 10, 25):
        S = w[0, k]
        _, Vxo, Vyo = om.pressure_step(S, q)
        Mw, Mo = om.rel_perm(S)
        L = ((Mw + Mo).reshape(om.shape) * om.K) ** (-1)
        TXo, TYo = np.zeros((n + 1, n)), np.zeros((n, n + 1))
        TXo[1:-1, :] = 2 * om.hy / om.hx / (L[0, :-1, :] + L[0, 1:, :])
        TYo[:, 1:-1] = 2 * om.hx / om.hy / (L[1, :, :-1] + L[1, :, 1:])
        x1, x2, y1, y2 = TXo[:-1].ravel(), TXo[1:].ravel(), TYo[:, :-1].ravel(), TYo[:, 1:].ravel()
        diag = y1 + y2 + x1 + x2
        diag[0] += np.sum(om.K[:, 0, 0])
        A = om.spdiags([-x2, -y2, diag, -y1, -x1], [-n, -1, 0, 1, n]).tocsc()
        P2 = spsolve(A, q, permc_spec="MMD_AT_PLUS_A").reshape(n, n)
        Vx2, Vy2 = np.zeros_like(Vxo), np.zeros_like(Vyo)
        Vx2[1:-1] = (P2[:-1] - P2[1:]) * TXo[1:-1]
        Vy2[:, 1:-1] = (P2[:, :-1] - P2[:, 1:]) * TYo[:, 1:-1]
        noise = max(np.abs(Vx2 - Vxo).max(), np.abs(Vy2 - Vyo).max())
        res_o = np.abs(((Vxo[1:] - Vxo[:-1]) + (Vyo[:, 1:] - Vyo[:, :-1])).ravel() - q).max()
        row = dict(step=k, noise=noise, res_oracle=res_o)
        for variant in (12, 15):
            pl = _plan(gm, 2, nTime=2)
            pl.set_variant(variant, 0)
            pl.set_inputs(perm_transf(x), transformed=True)
            pl.set_field("S", np.array([S, w[1, k]]))
            pl.pressure_only(0)
            Vx, Vy = pl.get_field("Vx")[0], pl.get_field("Vy")[0]
            pl.close()
            row[f"err{variant}"] = max(np.abs(Vx - Vxo).max(), np.abs(Vy - Vyo).max())
            row[f"res{variant}"] = np.abs(((Vx[1:] - Vx[:-1]) + (Vy[:, 1:] - Vy[:, :-1])).ravel() - q).max()
        rows.append(row)
        print({k2: (f"{v:.2e}" if isinstance(v, float) else v) for k2, v in row.items()})
    for row in rows:
        assert row["err12"] <= 10 * row["noise"] + 1e-12, row
        assert row["err15"] <= 10 * row["noise"] + 1e-12, row
        assert row["res12"] <= 10 * max(row["res_oracle"], row["noise"]) and row["res15"] <= 10 * max(row["res_oracle"], row["noise"]), row


def test_hand_over_to_the_cg_for_one_member():
    """The hand-over itself, forced (hm_fwd_set_debug "nd_force_fallback" = 1: member 1 of 3 is solved again by the two-level CG as a member block of one at every
    time step): clean status words, `nd_fallbacks` = the number of steps, the other members bit-identical to a run without any hand-over, member 1
    within the two solvers' rounding of its direct solve (both stop at rounding level on a well-conditioned member)."""
    n, N, steps = 256, 3, 5
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=12)
    out = []
    for force in (None, 1):
        plan = _plan(gm, N, nTime=steps)
        if force is not None:
            plan.set_debug("nd_force_fallback", force)
        plan.set_inputs(x, transformed=False)
        plan.run()
        st = plan.sync()
        w, p, status = plan.outputs()
        assert not status.any(), status
        assert st["nd_fallbacks"] == (0 if force is None else steps)
        out.append((w.copy(), plan.get_field("Vx").copy()))
        plan.close()
    (w0, v0), (w1, v1) = out
    for m in (0, 2):
        assert np.array_equal(w0[m], w1[m]) and np.array_equal(v0[m], v1[m])
    assert not np.array_equal(v0[1], v1[1])                      # a different solver did run
    assert np.abs(v0[1] - v1[1]).max() < 1e-7 and np.abs(w0[1] - w1[1]).max() < 1e-6


def test_nested_dissection_in_member_blocks_is_bit_identical():
    """An ensemble whose factor / update / panel buffers would exceed the memory budget is solved in blocks of members through the same
    buffers (hm_nd::cap; BASELINE config 4 whole on one GPU: 4096 members at 256 x 256).  Forced here with hm_fwd_set_debug "nd_cap" = 3 on 7 members: blocks
    of 3, 3 and 1 give what the whole ensemble in one block gives, bit for bit (nothing is kept across time steps in the blocked form)."""
    n, N, steps = 256, 7, 4
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=41)
    out = []
    for cap in (None, 3):
        plan = _plan(gm, N, nTime=steps)
        if cap is not None:
            plan.set_debug("nd_cap", cap)
        plan.set_inputs(x, transformed=False)
        plan.run()
        st = plan.sync()
        w, p, status = plan.outputs()
        assert not status.any() and st["mean_n_cg"] == 0
        out.append((w.copy(), p.copy(), plan.get_field("P").copy(), plan.get_field("Vx").copy()))
        plan.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_float32_slab_sweep_redo_launch_repairs_a_slab_that_sat_out():
    """The gated REDO launch of the float32 slab sweep (sat32s.hip: all_active = 2, HM_MEMBER_REDO_STEP) is what keeps slab skipping exact
    when water reaches a slab that sat the step out.  With the default rule (a slab sits out only if its neighbours are dry too) that
    happens to 1 member in 125 of config 5's prior; hm_fwd_set_debug "slab_margin" 0 lets a slab sit out as soon as it is dry itself, so the
    front crosses into a sitting-out slab within a few steps on EVERY member: slab_redos > 0, the status words clean (the REDO bit cleared),
    saturation history, producer series and sub-step counts bit-identical to the single-workgroup tiled sweep (sat_variant 3)."""
    nx, ny, N, steps = 512, 256, 6, 12
    _, gm = make_models(nx, ny, dtype=32)
    x = perms(nx, ny, N, seed=92)
    out = {}
    for mode in ("narrow", "tiled"):
        plan = _plan(gm, N, nTime=steps)
        if mode == "narrow":
            plan.set_debug("slab_margin", 0)
        else:
            plan.set_variant(0, 3)
        plan.set_inputs(x, transformed=False)
        plan.run(0, steps)
        st = plan.sync()
        w, p, status = plan.outputs()
        assert not status.any(), status  # (HM_MEMBER_REDO_STEP = 32 cleared by every redo)
        assert st["team_retries"] == 0
        out[mode] = (w, p, plan.get_field("nts").copy(), st["slab_redos"])
        plan.close()
    assert out["narrow"][3] > 0 and out["tiled"][3] == 0, out["narrow"][3]
    # the front did cross slab borders during the run: slabs 2 and 5 (rows 128..191, 320..383) were dry after the first step and are wet at the end
    S1, Send = out["narrow"][0][:, 1].reshape(N, nx, ny), out["narrow"][0][:, steps].reshape(N, nx, ny)
    assert (S1[:, 128:192] == 0).all() and (Send[:, 128:192] != 0).any() and (Send[:, 320:384] != 0).any()
    for a, b in zip(out["narrow"][:3], out["tiled"][:3]):
        assert np.array_equal(a, b), np.abs(a.astype(float) - b.astype(float)).max()


@pytest.mark.gpu
def test_embedded_fp32_plan_sees_a_state_written_through_the_raw_pointer():
    """hm_fwd_device_ptr("S") hands out the OUTER plan's saturation; an embedded grid (200 x 150 inside 256 x 256, dtype = 32) must then
    keep no record of wet slabs in its INNER plan either (forward.hip: embedded_inner passes raw_state_exposed on): water written through the
    pointer into a slab that sat the previous steps out is swept, not zeroed.  Against the generic kernels on the grid as given ("embed" 0):
    within the float32 bar, the written water still there."""
    import ctypes as C

    from historymatching_amd import _lib

    nx, ny, N, steps = 200, 150, 3, 3
    _, gm = make_models(nx, ny, dtype=32)
    x = perms(nx, ny, N, seed=93)
    res = {}
    for name in ("embedded", "generic"):
        plan = _plan(gm, N, nTime=steps + 2, keep_history=False)
        if name == "generic":
            plan.set_debug("embed", 0)
        plan.set_inputs(x, transformed=False)
        plan.run(0, steps)
        plan.sync()
        S = np.ascontiguousarray(plan.get_field("S").reshape(N, nx, ny))
        assert (S[:, 192:] == 0).all()  # rows 192.. (the inner plan's last slab of 64 rows) are dry: that slab sat out
        S[:, 193:199, 20:60] = np.float32(0.3)
        ptr = plan.device_ptr("S")
        _lib.check(plan.lib.hm_copy_to_device(plan.ctx.handle, C.c_void_p(ptr), S.ctypes.data_as(C.c_void_p), S.nbytes), "copy S")
        plan.run(steps, 2)
        plan.sync()
        _, _, status = plan.outputs(want_wsats=False)
        assert not status.any()
        res[name] = plan.get_field("S").reshape(N, nx, ny).astype(np.float64)
        plan.close()
    assert res["embedded"][:, 193:199, 20:60].min() > 0.05  # the water written by hand was swept, not zeroed
    assert np.abs(res["embedded"] - res["generic"]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("teams", [2, 4])
def test_small_shard_team_sweep_128_is_bit_identical_to_the_one_workgroup_sweep(teams):
    """Small member shards at 128 x 128 (one rank's share of a strong-scaled ensemble: members x slabs <= CUs) run the fp64 sweep as TEAMS
    of two or four slab workgroups per member (sat128s.hip), each on a CU of its own, instead of sat128r's one workgroup per member.  Same
    arithmetic per cell: saturation history, producer series and sub-step counts of a 12-step run equal the one-workgroup sweep's
    (hm_fwd_set_debug "sat_teams" 0) bit for bit -- with the default automatic choice as well (24 members: teams of four; 100: of two) --
    and nobody waited in vain (team_retries 0).  Also from a state with water in every band (late-run: every slab border is crossed)."""
    n, steps = 128, 12
    _, gm = make_models(n, n)
    for N, auto in ((24, 4), (100, 2)):
        if teams * ((N + 7) // 8) * 8 > 256:  # (teams of four need 4 x 104 CUs for 100 members: not on this chip)
            continue
        x = perms(n, n, N, seed=95 + N)
        out = {}
        for mode in ("one", "teams", "auto"):
            if mode == "auto" and auto != teams:
                continue
            plan = _plan(gm, N, nTime=steps + 2)
            plan.set_debug("sat_teams", {"one": 0, "teams": teams, "auto": -1}[mode])
            plan.set_inputs(x, transformed=False)
            plan.run(0, steps)
            S = plan.get_field("S")
            S[:, ::7, 3::11] = 0.25  # water in every band of 16 rows
            plan.set_field("S", S)
            plan.run(steps, 2)
            st = plan.sync()
            w, p, status = plan.outputs()
            assert not status.any() and st["team_retries"] == 0
            out[mode] = (w, p, plan.get_field("nts").copy(), st["ms_saturation"] / st["n_saturation_launches"])
            plan.close()
        assert len(out) >= 2
        for mode in out:
            if mode == "one":
                continue
            for a, b in zip(out["one"][:3], out[mode][:3]):
                assert np.array_equal(a, b), (N, mode, np.abs(a - b).max())
            assert out[mode][3] < out["one"][3], (N, mode, out[mode][3], out["one"][3])  # and it is the faster one at this shard size


@pytest.mark.gpu
def test_small_shard_pressure_levels_one_front_per_workgroup_is_bit_identical():
    """A shard of fewer members than CUs eliminates levels 3 .. 0 of the 128 x 128 nested dissection as a launch per level, one front per
    workgroup (press_nd.hip: the larger grids' form of k_nd_top), instead of one workgroup per member taking the 15 fronts in turn.  Same
    tiles, same products: pressures, fluxes and the saturation after four steps equal the member-per-workgroup form's
    (hm_fwd_set_debug "top_per_level" 0) bit for bit, with and without the reuse of dry fronts."""
    n, N, steps = 128, 20, 4
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=97)
    out = {}
    for per_level in (0, 1):
        for variant in (0, 14):
            plan = _plan(gm, N, nTime=steps, keep_history=False)
            plan.set_variant(variant, 0)
            plan.set_debug("top_per_level", per_level)
            plan.set_inputs(x, transformed=False)
            plan.run(0, steps)
            plan.sync()
            S, _, status = plan.outputs()
            assert not status.any()
            out[per_level, variant] = (plan.get_field("P").copy(), plan.get_field("Vx").copy(), plan.get_field("Vy").copy(), S)
            plan.close()
    for key in ((1, 0), (0, 14), (1, 14)):
        for a, b in zip(out[0, 0], out[key]):
            assert np.array_equal(a, b), key


@pytest.mark.gpu
@pytest.mark.parametrize("n,N", [(128, 300), (128, 20), (256, 12)])
def test_top_front_tile_dealing_and_trickled_prefetch_are_bit_identical(n, N):
    """Round 6, k_nd_top: the trailing tiles of a front are dealt in row-major runs to the waves that own no pivot tile, the next front's tables
    and children arrive piece by piece behind the tile updates, and the update's stores drain beside the next front's gathers
    (hm_fwd_set_debug "top_deal" 1, the default).  Which wave holds a tile and when a copy is issued change no arithmetic: pressures, fluxes
    and the saturation after four steps equal those of rounds 3-5's form ("top_deal" 0) bit for bit -- a member per workgroup (300 members),
    a front per workgroup (20 members: a launch per level) and the larger grids' instances of the kernel (256 x 256), with and without the
    reuse of dry fronts."""
    steps = 4
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=131)
    out = {}
    for deal in (0, 1):
        for variant in (0, 14):
            plan = _plan(gm, N, nTime=steps, keep_history=False)
            plan.set_variant(variant, 0)
            plan.set_debug("top_deal", deal)
            plan.set_inputs(x, transformed=False)
            plan.run(0, steps)
            plan.sync()
            S, _, status = plan.outputs()
            assert not status.any()
            out[deal, variant] = (plan.get_field("P").copy(), plan.get_field("Vx").copy(), plan.get_field("Vy").copy(), S)
            plan.close()
    for key in ((1, 0), (0, 14), (1, 14)):
        for a, b in zip(out[0, 0], out[key]):
            assert np.array_equal(a, b), key


@pytest.mark.gpu
def test_lazy_face_fluxes_are_bit_identical_and_materialised_on_demand():
    """Round 6: at 128 x 128 the pressure step leaves P, TX, TY and launches no flux kernel; the default sweep forms the fluxes of its patch
    from them (the same expression), and Vx / Vy appear when somebody asks (hm_fwd_get_field).  Against hm_fwd_set_debug "lazy_flux" 0 over a
    run with fields read at several steps, a pressure-only / saturation-only sequence and fluxes GIVEN by the caller: bit-identical."""
    n, N, steps = 128, 12, 6
    _, gm = make_models(n, n)
    x = perms(n, n, N, seed=101)
    out = {}
    for lazy in (0, 1):
        plan = _plan(gm, N, nTime=steps + 2, keep_history=False)
        plan.set_debug("lazy_flux", lazy)
        plan.set_inputs(x, transformed=False)
        res = []
        plan.run(0, 3)
        res += [plan.get_field("Vx").copy(), plan.get_field("Vy").copy(), plan.get_field("S").copy()]  # fluxes of step 2, asked for after its sweep
        plan.run(3, 3)
        plan.pressure_only(6)
        res += [plan.get_field("P").copy(), plan.get_field("Vx").copy()]
        plan.saturation_only(6)
        res += [plan.get_field("S").copy(), plan.get_field("nts").copy()]
        Vx = plan.get_field("Vx")
        plan.set_field("Vx", 0.5 * Vx)          # the caller's fluxes: the sweep must take these, not recompute from P
        plan.set_field("Vy", 0.5 * plan.get_field("Vy"))
        plan.saturation_only(7)
        res += [plan.get_field("S").copy(), plan.get_field("nts").copy()]
        _, _, status = plan.outputs()
        assert not status.any()
        out[lazy] = res
        plan.close()
    assert not np.array_equal(out[1][-2], out[1][-4])  # (the halved fluxes did change the step)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
