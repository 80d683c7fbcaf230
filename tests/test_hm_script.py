"""BASELINE config 1 -- "Reference HistoryMatch.py CPU run" -- replayed from fixture F9 (tests/golden/f9_hm_script.npz: what the
reference's own script computes when it runs end to end on the oracle simulator, captured by oracle/make_golden_script.py).

CPU part: the oracle reproduces the fixture (the simulator half bit for bit: it produced it; the update half to rounding: the fixture's
posteriors are the REFERENCE's arithmetic).  GPU part (-m gpu): the same workflow through the drop-in of INTEGRATION.md section 1 --
`ressim.ResSim` for `TPFA_ResSim`, `make_forward_model` for `forward_model`, `ens_update0 / ens_update0_loc / ies / iles` for the
notebook's own functions -- with the notebook's model, wells, truth, prior and random numbers.  HistoryMatch.py lines in the tests."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def f9(golden):
    return np.load(golden / "f9_hm_script.npz")


def _oracle_model(f9):
    from oracle.ressim import ResSim

    m = ResSim(int(f9["Nx"]), int(f9["Ny"]), float(f9["Lx"]), float(f9["Ly"]))   # HistoryMatch.py:97
    m.inj_xy, m.prd_xy = f9["inj_xy"], f9["prd_xy"]                                # :177-190
    m.inj_rates, m.prd_rates = f9["inj_rates"], f9["prd_rates"]
    return m


def _gpu_model(f9):
    from historymatching_amd.ressim import ResSim

    m = ResSim(int(f9["Nx"]), int(f9["Ny"]), float(f9["Lx"]), float(f9["Ly"]))
    m.inj_xy, m.prd_xy = f9["inj_xy"], f9["prd_xy"]
    m.inj_rates, m.prd_rates = f9["inj_rates"], f9["prd_rates"]
    return m


def test_fixture_is_the_reference_configuration(f9):
    # HistoryMatch.py:97, 219-221, 289: 20 x 20 cells on 2 x 1, T = 1 in 40 steps, N = 40 members; 4 producers x 40 times = 160 observations
    assert (int(f9["Nx"]), int(f9["Ny"]), float(f9["Lx"]), float(f9["Ly"])) == (20, 20, 2.0, 1.0)
    assert float(f9["dt"]) == 0.025 and int(f9["nTime"]) == 40
    assert f9["perm_Prior"].shape == (40, 400) and f9["prod_past_Prior"].shape == (40, 40, 4) and f9["obs"].shape == (160,)
    # the seed-1 random stream of the script -- the same anchor values fixture F1 holds
    assert np.allclose(f9["perm_Truth"][0, :3], [1.62434536, 1.51231157, 1.35084072])
    # the script's own claim (HistoryMatch.py:1187-1196): every method brings the ensemble mean closer to the truth than the prior's
    rms = lambda E: float(np.sqrt(np.mean((E.mean(0) - f9["perm_Truth"][0]) ** 2)))  # noqa: E731
    assert max(rms(f9[k]) for k in ("perm_ES", "perm_LES", "perm_IES", "perm_ILES")) < rms(f9["perm_Prior"])


def test_oracle_simulator_reproduces_the_scripts_runs(f9):
    """Truth run (HistoryMatch.py:224-225) and prior ensemble run (:400-401) of the script, from the oracle called directly."""
    from oracle.ressim import forward_model, set_perm

    om = _oracle_model(f9)
    set_perm(om, f9["perm_Truth"][0])
    w = om.sim(float(f9["dt"]), int(f9["nTime"]), np.zeros(om.Nxy))
    assert np.array_equal(w[1:], f9["wsat_past_Truth"]) or np.array_equal(w, f9["wsat_past_Truth"])
    assert np.array_equal(w[1:, f9["prod_inds"]], f9["prod_past_Truth"])
    ws, ps = forward_model(om, f9["perm_Prior"][:6], None, float(f9["dt"]), int(f9["nTime"]))
    assert np.array_equal(ps, f9["prod_past_Prior"][:6]) and np.array_equal(ws[:, -1], f9["wsat_final_Prior"][:6])


def test_oracle_updates_reproduce_the_scripts_posteriors(f9):
    """perm.ES (HistoryMatch.py:652) and perm.LES (:863) from the fixture's own inputs through oracle/es.py."""
    from oracle import es

    kw = dict(obs_ens=f9["obs_ens"], obs=f9["obs"], perturbs=f9["perturbs"], decorr=f9["decorr"])
    assert np.array_equal(f9["obs_ens"], es.vect(f9["prod_past_Prior"], 40))           # hm_setup0, :635-640
    assert np.abs(es.ens_update0(f9["perm_Prior"], **kw) - f9["perm_ES"]).max() < 1e-10
    assert np.abs(es.ens_update0_loc(f9["perm_Prior"], **kw, taper=f9["taper_LES"]) - f9["perm_LES"]).max() < 1e-10


def test_product_host_helpers_on_the_scripts_data(f9):
    from historymatching_amd import obs as pobs
    from historymatching_amd.localization import taper_for_wells

    assert np.array_equal(pobs.vect(f9["prod_past_Prior"], 40), f9["obs_ens"])
    R, R12 = pobs.obs_error_model(40, 4)[:2]
    assert np.abs(pobs.decorr(R12) - f9["decorr"]).max() < 1e-12
    taper = taper_for_wells(_oracle_model(f9), f9["prod_inds"], 40, radius=1.2)        # :700-717, 863
    assert np.abs(taper - f9["taper_LES"]).max() < 1e-14


# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_config1_forward_runs_through_the_drop_in(f9):
    """`model.sim` for the truth (HistoryMatch.py:224) and `forward_model(perm.Prior)` (:400-401) on the GPU against the script's arrays:
    same sub-step counts, saturations within the oracle's own solver noise (tests/helpers.py), producer series = the gather of :212-213."""
    from historymatching_amd.forward import make_forward_model, perm_transf
    from oracle.ressim import set_perm
    from tests.helpers import oracle_sim_and_noise

    gm, om = _gpu_model(f9), _oracle_model(f9)
    dt, nTime = float(f9["dt"]), int(f9["nTime"])
    gm.K = np.stack([perm_transf(f9["perm_Truth"][0]).reshape(gm.shape)] * 2)          # set_perm, :160-164
    w = gm.sim(dt, nTime, np.zeros(gm.Nxy))
    ref, noise = oracle_sim_and_noise(om, f9["perm_Truth"][0], dt, nTime)
    assert w.shape == (nTime + 1, gm.Nxy) and np.array_equal(w[0], np.zeros(gm.Nxy))
    assert np.abs(w - ref).max() <= 10 * noise + 1e-9
    assert np.abs(w[1:][:, f9["prod_inds"]] - f9["prod_past_Truth"]).max() <= 10 * noise + 1e-9
    forward_model = make_forward_model(gm, dt, nTime)
    wsats, prods = forward_model(f9["perm_Prior"])                                      # :400-401
    assert wsats.shape == (40, nTime + 1, 400) and prods.shape == (40, nTime, 4)
    assert np.array_equal(prods, wsats[:, 1:, :][:, :, f9["prod_inds"]])
    worst = 0.0
    for m in range(0, 40, 5):
        ref, noise = oracle_sim_and_noise(om, f9["perm_Prior"][m], dt, nTime)
        assert np.array_equal(ref[1:, f9["prod_inds"]], f9["prod_past_Prior"][m])     # (the fixture IS the oracle's run)
        err = np.abs(wsats[m] - ref).max()
        assert err <= 10 * noise + 1e-9, (m, err, noise)
        worst = max(worst, err)
    assert np.abs(prods - f9["prod_past_Prior"]).max() < 1e-6 and worst < 1e-6
    forward_model.release()


@pytest.mark.gpu
def test_config1_updates_through_the_drop_in(f9):
    """ES (HistoryMatch.py:652), localised ES (:863), IES (:958-961) and ILES (:1075-1077) of the script through the product's functions
    on the fixture's own inputs, <= 1e-10 of the reference's posteriors.  The iterative smoothers call the forward model once per
    iterate (:921, :1027): here the observation function replays the simulated observations the reference's iterates produced (stored in
    the fixture), so what is compared is the update arithmetic alone, iterate by iterate."""
    from historymatching_amd.update import ens_update0, ens_update0_loc, ies, iles

    kw = dict(obs=f9["obs"], perturbs=f9["perturbs"], decorr=f9["decorr"])
    assert np.abs(ens_update0(f9["perm_Prior"], obs_ens=f9["obs_ens"], **kw) - f9["perm_ES"]).max() < 1e-10
    assert np.abs(ens_update0_loc(f9["perm_Prior"], obs_ens=f9["obs_ens"], **kw, taper=f9["taper_LES"]) - f9["perm_LES"]).max() < 1e-10

    def replay(series):
        it = iter(series)
        return lambda E: next(it)

    xStep, iMax = float(f9["IES_xStep"]), int(f9["IES_iMax"])
    for subspace in ("gram", "svd", "device"):
        post, stats = ies(f9["perm_Prior"], replay(f9["IES_Eo"]), **kw, xStep=xStep, iMax=iMax, subspace=subspace)
        assert np.abs(post - f9["perm_IES"]).max() < 1e-9, subspace   # ten Gauss-Newton iterates deep: rounding grows with the iterate count
        assert len(stats["E"]) == iMax
    post, stats = iles(f9["perm_Prior"], replay(f9["ILES_Eo"]), **kw, taper=f9["taper_LES"], xStep=xStep, iMax=iMax)
    assert np.abs(post - f9["perm_ILES"]).max() < 1e-9


@pytest.mark.gpu
def test_config1_iterative_smoother_with_the_gpu_forward_model(f9):
    """The IES of the script (HistoryMatch.py:958-961) with the GPU forward model inside the loop, as the notebook runs it: ten iterates, each a
    40-member ensemble run.  The simulator's rounding-level differences pass through ten Gauss-Newton steps; the posterior stays within
    1e-5 of the reference's and reduces the error like it."""
    from historymatching_amd.forward import make_forward_model
    from historymatching_amd.obs import vect
    from historymatching_amd.update import ies

    gm = _gpu_model(f9)
    forward_model = make_forward_model(gm, float(f9["dt"]), int(f9["nTime"]))
    obs_fun = lambda x: vect(forward_model(x)[1], 40)  # noqa: E731  (hm_setupI, :958-959)
    post, stats = ies(f9["perm_Prior"], obs_fun, obs=f9["obs"], perturbs=f9["perturbs"], decorr=f9["decorr"],
                      xStep=float(f9["IES_xStep"]), iMax=int(f9["IES_iMax"]))
    forward_model.release()
    assert np.abs(stats["Eo"][0] - f9["IES_Eo"][0]).max() < 1e-6
    assert np.abs(post - f9["perm_IES"]).max() < 1e-5
    rms = lambda E: float(np.sqrt(np.mean((E.mean(0) - f9["perm_Truth"][0]) ** 2)))  # noqa: E731
    assert rms(post) < rms(f9["perm_Prior"])
