"""Pins oracle/es.py against fixtures captured from the reference's own functions (oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import es


def _load(golden, name):
    return np.load(golden / name)


def test_rng_replay_anchor_values(golden):
    f1 = _load(golden, "f1_rng_replay.npz")
    # first values of perm.Truth, prior mean/var
    assert np.allclose(f1["perm_truth"][0, :3], [1.62434536, 1.51231157, 1.35084072])
    assert abs(f1["perm_prior"].mean() - 0.0607) < 1e-4 and abs(f1["perm_prior"].var() - 1.0118) < 1e-4
    assert f1["perm_prior"].shape == (40, 400) and f1["hm_perturbs"].shape == (40, 160)


def test_obs_error_model(golden):
    f2 = _load(golden, "f2_obs_error.npz")
    R, R12, decorr = es.obs_error_model(40, 4)
    assert np.array_equal(R, f2["R"])
    assert np.allclose(R12, f2["R12"], rtol=0, atol=1e-15)
    assert np.allclose(decorr, f2["decorr"], rtol=0, atol=1e-12)


def test_ens_update0_gaussian_gaussian_bugcheck(golden):
    """HistoryMatch.py:594-612: posterior ~ N(y/4 = 1, I) up to sampling error."""
    f1, f3 = _load(golden, "f1_rng_replay.npz"), _load(golden, "f3_ens_update0.npz")
    E = f1["gg_E"]
    post = es.ens_update0(E, E, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3))
    assert np.abs(post - f3["gg_postr"]).max() < 1e-13
    assert np.allclose(post.mean(0), [0.98474734, 1.08434752, 1.02224568], atol=1e-8)
    assert np.abs(np.cov(post.T) - np.eye(3)).max() < 0.15


def test_ens_update0_history_matching_shapes(golden):
    f1, f2, f3 = (_load(golden, n) for n in ("f1_rng_replay.npz", "f2_obs_error.npz", "f3_ens_update0.npz"))
    kw = dict(obs_ens=f3["obs_ens"], obs=f3["obs"], perturbs=f1["hm_perturbs"], decorr=f2["decorr"])
    assert np.abs(es.ens_update0(f1["perm_prior"], **kw) - f3["perm_es"]).max() < 1e-12
    assert np.abs(es.ens_update0(f3["obs_ens"], **kw) - f3["es0"]).max() < 1e-12  # HistoryMatch.py:1156


def test_ens_update0_loc(golden):
    f1, f2, f3, f4 = (_load(golden, n) for n in ("f1_rng_replay.npz", "f2_obs_error.npz", "f3_ens_update0.npz", "f4_ens_update0_loc.npz"))
    E = f1["gg_E"]
    gg = es.ens_update0_loc(E, E, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), np.eye(3))
    assert np.abs(gg - f4["gg_postr_loc"]).max() < 1e-13
    kw = dict(obs_ens=f3["obs_ens"], obs=f3["obs"], perturbs=f1["hm_perturbs"], decorr=f2["decorr"])
    ones = es.ens_update0_loc(f1["perm_prior"], **kw, taper=np.ones((400, 160)))
    assert np.abs(ones - f4["les_ones"]).max() < 1e-12
    assert np.allclose(ones, f3["perm_es"])  # "Reproduces global analysis?" HistoryMatch.py:821-822
    les = es.ens_update0_loc(f1["perm_prior"], **kw, taper=f4["taper"])
    assert np.abs(les - f4["perm_les"]).max() < 1e-12


def test_helpers(golden):
    f5 = _load(golden, "f5_helpers.npz")
    X, x = es.center(f5["a"])
    assert np.array_equal(X, f5["center_X"]) and np.array_equal(x, f5["center_x"])
    assert np.array_equal(es.center(f5["a"], rescale=True)[0], f5["center_Xr"])
    assert np.allclose(es.cov(f5["a"], f5["b"]), f5["cov"], rtol=0, atol=1e-15)
    assert np.allclose(es.corr(f5["a"], f5["b"][:, 0]), f5["corr"], rtol=0, atol=1e-15)
    for s, ref in zip(f5["bump_sharp"], f5["bumps"]):
        assert np.array_equal(es.bump(f5["bump_x"], s), ref)
    assert np.array_equal(es.pairwise_distances(f5["pd_A"]), f5["pd_AA"])
    assert np.array_equal(es.pairwise_distances(np.arange(4)[:, None], [[2]]), f5["pd_1d"])
    assert np.array_equal(es.pairwise_distances(np.arange(4)[:, None], domain=(4,)), f5["pd_periodic"])
    # doctest values of the reference (localization.py:31-60, geostat.py:19-22)
    assert np.allclose(f5["pd_AA"][0], [0, 1, 1, 2**0.5])
    assert np.allclose(f5["vg"], [0.0, 0.6689085, 0.98351593])


def test_taper_from_restated_grid(golden):
    """Taper wiring HistoryMatch.py:700-717, 863 on the restated 20x20 grid."""
    from oracle.ressim import ResSim, default_wells

    f4 = _load(golden, "f4_ens_update0_loc.npz")
    model = default_wells(ResSim(20, 20, 2, 1))
    prod_inds = model.xy2ind(*model.prd_xy.T)
    assert np.array_equal(prod_inds, f4["prod_inds"])
    xy_obs = np.tile(model.ind2xy(prod_inds), 40)
    xy_prm = model.ind2xy(np.arange(model.Nxy))
    d = es.pairwise_distances(xy_prm.T, xy_obs.T)
    assert np.array_equal(d, f4["distances_to_obs"])
    assert np.array_equal(es.bump(d / 1.2), f4["taper"])


def test_product_localization_helpers_match_reference_fixtures(golden):
    """The PRODUCT's taper construction (historymatching_amd/localization.py: bump, pairwise_distances, taper_for_wells;
    inputs of ens_update0_loc) against the values captured from the reference's own functions: F5 (localization.py:31-60
    doctests, bump for the six sharpness values of HistoryMatch.py:687-690) and F4 (the wiring of HistoryMatch.py:700-717,
    863 on the 20x20 grid).  No GPU: only the grid conventions of the GPU-backed model are touched."""
    from historymatching_amd import localization as loc
    from historymatching_amd.ressim import ResSim
    from tests.helpers import wells_4corners

    f5 = _load(golden, "f5_helpers.npz")
    for s, ref in zip(f5["bump_sharp"], f5["bumps"]):
        assert np.array_equal(loc.bump(f5["bump_x"], s), ref)
    assert np.array_equal(loc.pairwise_distances(f5["pd_A"]), f5["pd_AA"])
    assert np.array_equal(loc.pairwise_distances(np.arange(4)[:, None], [[2]]), f5["pd_1d"])
    assert np.array_equal(loc.pairwise_distances(np.arange(4)[:, None], domain=(4,)), f5["pd_periodic"])
    f4 = _load(golden, "f4_ens_update0_loc.npz")
    gm = wells_4corners(ResSim(20, 20, 2, 1))
    prod_inds = gm.xy2ind(*gm.prd_xy.T)
    assert np.array_equal(prod_inds, f4["prod_inds"])
    xy_obs = np.tile(gm.ind2xy(prod_inds), 40)
    assert np.array_equal(loc.pairwise_distances(gm.ind2xy(np.arange(gm.Nxy)).T, xy_obs.T), f4["distances_to_obs"])
    assert np.array_equal(loc.taper_for_wells(gm, prod_inds, 40, radius=1.2), f4["taper"])


def test_product_obs_helpers_match_reference_fixture(golden):
    """historymatching_amd.obs (host-side inputs of the update) against the fixture captured from
    the reference's own construction (HistoryMatch.py:243-259, 639) and its seed-1 RNG replay (:600-603)."""
    from historymatching_amd import obs as pobs

    f1, f2 = _load(golden, "f1_rng_replay.npz"), _load(golden, "f2_obs_error.npz")
    R, R12 = pobs.obs_error_model(40, 4)
    assert np.array_equal(R, f2["R"]) and np.array_equal(R12, f2["R12"])
    assert np.abs(pobs.decorr(R12) - f2["decorr"]).max() < 1e-12
    x = np.arange(2 * 40 * 4.0).reshape(2, 40, 4)
    assert np.array_equal(pobs.vect(pobs.vect(x), 40, undo=True), x) and pobs.vect(x).shape == (2, 160)
    rng = np.random.RandomState(3)
    z = rng.randn(5, 160)
    rng = np.random.RandomState(3)
    assert np.array_equal(pobs.perturbations(5, R12, rng), z @ R12.T)


def test_product_iles_matches_reference_fixture(golden):
    """historymatching_amd.update.iles_host (host algebra; the device path is tested in test_update_gpu.py) against the output of the REAL reference `ILES` on
    the linear-Gaussian bug check (fixture F6 `iles_gg`), which must also reproduce the non-iterative local analysis
    (HistoryMatch.py:1069-1071)."""
    from historymatching_amd.update import iles_host as iles

    f1, f4, f6 = _load(golden, "f1_rng_replay.npz"), _load(golden, "f4_ens_update0_loc.npz"), _load(golden, "f6_iterative.npz")
    post, stats = iles(f1["gg_E"], lambda x: x, 4 * np.ones(3), f1["gg_perturbs"], 0.5 * np.eye(3), taper=np.eye(3))
    assert len(stats["E"]) == 4
    assert np.abs(post - f6["iles_gg"]).max() < 1e-9
    assert np.allclose(post, f4["gg_postr_loc"])


def test_npv_accounting_and_member_validation_host_logic():
    """historymatching_amd/opt.py host side (no GPU): the accounting equals the oracle's restatement of
    Optimise.py:170-200 on the same saturations, and invalid member configurations are recognised the way the reference's
    npv() penalises them (Optimise.py:119-124)."""
    from historymatching_amd import opt
    from historymatching_amd.ressim import ResSim
    from oracle import opt as oopt
    from oracle.ressim import ResSim as OResSim

    n, dt, nT = 12, 0.025, 6
    om = OResSim(n, n, 2, 1)
    om.K = 0.1 + np.exp(np.random.RandomState(3).randn(n * n))
    om.inj_xy, om.prd_xy = [[1.0, 0.5]], [[0.2, 0.2], [1.8, 0.8]]
    om.inj_rates, om.prd_rates = 1.5 * np.ones((1, 1)), 0.75 * np.ones((2, 1))
    value, wsats = oopt.npv(om, dt, nT, np.zeros(n * n))
    s = wsats[:, om.xy2ind(*om.prd_xy.T)]
    ledger = opt.accounting(((s[:-1] + s[1:]) / 2).T, np.broadcast_to(om.inj_rates, (1, nT)), np.broadcast_to(om.prd_rates, (2, nT)),
                            dt, opt.default_prices(dt), opt.discounts(dt, nT), 1.5)
    assert abs(sum(ledger.values()) - value) <= 1e-12 * abs(value)
    gm = ResSim.__new__(ResSim)  # grid conventions only: no device is touched
    gm.__dict__.update(Nx=n, Ny=n, Lx=2.0, Ly=1.0)
    gm._inj_xy, gm._prd_xy = np.array([[1.0, 0.5]]), np.array([[0.2, 0.2], [1.8, 0.8]])
    gm.inj_rates, gm.prd_rates = om.inj_rates, om.prd_rates
    inj_ind, inj, prd_ind, prd = opt._member_config(gm, {"inj_xy": [[0.4, 0.9]]}, nT)
    assert inj_ind[0] == om.xy2ind(0.4, 0.9) and inj.shape == (1, nT) and list(prd_ind) == list(om.xy2ind(*om.prd_xy.T))
    for bad in ({"inj_xy": [[2.2, 0.5]]}, {"prd_rates": np.ones((2, 1))}, {"inj_rates": np.ones((1, 3))}):
        with pytest.raises(ValueError):
            opt._member_config(gm, bad, nT)


def test_kronecker_prior_has_the_reference_covariance():
    """The reference samples the prior from the dense covariance 1 - variogram_gauss(pairwise distances) of all cell centres
    (notebooks/tools/geostat.py:86-99).  With the Gaussian variogram that matrix is exactly Cx (x) Cy in the C-order cell
    index ix*Ny + iy, which is what historymatching_amd.geostat samples per axis; the sampler's factors reproduce it."""
    from historymatching_amd.geostat import _axis_factors, variogram_gauss

    Nx, Ny, Lx, Ly, r = 7, 5, 2.0, 1.0, 0.8
    xc, yc = (np.arange(Nx) + 0.5) * Lx / Nx, (np.arange(Ny) + 0.5) * Ly / Ny
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel()], 1)
    dense = 1 - variogram_gauss(es.pairwise_distances(pts), r)
    Cx = 1 - variogram_gauss(np.abs(xc[:, None] - xc), r)
    Cy = 1 - variogram_gauss(np.abs(yc[:, None] - yc), r)
    assert np.abs(dense - np.kron(Cx, Cy)).max() < 1e-14
    Ux, Uy = _axis_factors(Nx, Ny, Lx, Ly, r)
    # Cov of vec(Ux^T Z Uy) = (Ux^T Ux) (x) (Uy^T Uy)
    assert np.abs(np.kron(Ux.T @ Ux, Uy.T @ Uy) - dense).max() < 1e-9


def test_rectangular_partitioning_matches_reference_fixture():
    """historymatching_amd.localization.rectangular_partitioning against outputs of the reference helper
    (notebooks/tools/localization.py:95-145; fixture F7, oracle/make_golden_partitioning.py)."""
    from pathlib import Path

    from historymatching_amd.localization import rectangular_partitioning

    f7 = np.load(Path(__file__).parent / "golden" / "f7_partitioning.npz")
    c = 0
    while f"c{c}_shape" in f7:
        shape, steps = list(f7[f"c{c}_shape"]), list(f7[f"c{c}_steps"])
        batches = rectangular_partitioning(shape, steps)
        assert [len(b) for b in batches] == list(f7[f"c{c}_len"])
        assert np.array_equal(np.concatenate(batches), f7[f"c{c}_ind"])
        sub = rectangular_partitioning(shape, steps, do_ind=False)
        assert np.array_equal(np.concatenate([np.stack(b, 0) for b in sub], axis=1), f7[f"c{c}_sub"])
        assert sorted(np.concatenate(batches)) == list(range(int(np.prod(shape))))  # a partition of the grid
        c += 1
    assert c == 5


def test_npv_accounting_matches_the_reference_fixture(golden):
    """F10 (oracle/make_golden_npv.py): the ledgers of the reference's OWN `accounting` / `prd_sats` (Optimise.py:170-208, AST-extracted) on
    six synthetic cases -- shut-in intervals, a late producer, field production above rate0, changing injection rates.  Both the oracle's
    restatement and the product's accounting reproduce every entry."""
    from historymatching_amd import opt as popt
    from oracle import opt as oopt

    f = _load(golden, "f10_npv_accounting.npz")
    dt, nTime, rate0 = float(f["dt"]), int(f["nTime"]), float(f["rate0"])
    keys = [str(k) for k in f["ledger_keys"]]
    price = dict(zip((str(k) for k in f["price_keys"]), f["price_values"]))
    assert popt.default_prices(dt) == pytest.approx(price) and np.allclose(popt.discounts(dt, nTime), f["discounts"], rtol=0, atol=1e-15)
    for i in range(int(f["n_cases"])):
        wsats, cells, inj, prd, ref = f[f"wsats_{i}"], f[f"prd_cells_{i}"], f[f"inj_{i}"], f[f"prd_{i}"], f[f"ledger_{i}"]
        s = wsats[:, cells]
        prd_wsats = ((s[:-1] + s[1:]) / 2).T   # prd_sats, Optimise.py:205-208 (the product forms it the same way: opt.py NpvBatch.__call__)
        assert np.array_equal(prd_wsats, f[f"prd_wsats_{i}"])
        lo = oopt.accounting(prd_wsats, inj, prd, dt, nTime, rate0)
        lp = popt.accounting(prd_wsats, inj, prd, dt, popt.default_prices(dt), popt.discounts(dt, nTime), rate0)
        for k, r in zip(keys, ref):
            assert abs(lo[k] - r) <= 1e-12 * max(1.0, abs(r)), (i, k, lo[k], r)
            assert abs(lp[k] - r) <= 1e-12 * max(1.0, abs(r)), (i, k, lp[k], r)


def test_prior_law_matches_the_reference_fixture(golden):
    """F11 (oracle/make_golden_prior.py): rows of the reference's dense prior covariance on the default 20 x 20 grid, formed by the
    reference's own vectorize / dist_euclid / variogram_gauss (notebooks/tools/geostat.py:33-47, 10-30, 86-99).  The oracle's dense
    restatement reproduces them, and so does the covariance the product's per-axis (Kronecker) sampler has by construction."""
    from historymatching_amd.geostat import _axis_factors
    from oracle import geostat as og

    f = _load(golden, "f11_prior_law.npz")
    Nx, Ny, Lx, Ly, r = int(f["Nx"]), int(f["Ny"]), float(f["Lx"]), float(f["Ly"]), float(f["r"])
    Cov = og.covariance(Nx, Ny, Lx, Ly, r)
    assert np.abs(Cov[f["cells"]] - f["cov_rows"]).max() < 1e-15 and np.abs(np.diag(Cov) - f["cov_diag"]).max() < 1e-15
    Ux, Uy = _axis_factors(Nx, Ny, Lx, Ly, r)
    assert np.abs(np.kron(Ux.T @ Ux, Uy.T @ Uy)[f["cells"]] - f["cov_rows"]).max() < 1e-9   # (the 1e-10 nugget per axis)
    # the oracle's sampler draws what the reference's draws from the same normals (same factorisation of the same matrix)
    ref3 = f["ref_sample_fields_first3"]
    mine = og.gaussian_fields(Nx, Ny, Lx, Ly, 2000, r=r, rng=np.random.RandomState(7))[:3]
    assert np.abs(mine - ref3).max() < 1e-6   # Cov is numerically singular: its Cholesky factor is only reproducible to ~1e-8
