"""Generate tests/golden/*.npz from the REAL reference functions.  Runs only where /root/reference exists
(this build container); the GPU box and the test-suite only ever read the committed .npz files.

What is executed from the reference (imported / AST-extracted, never copied into the repo):
  * notebooks/tools/utils.py         center, cov, corr            (stub for the plotting-only `adjustText`)
  * notebooks/tools/localization.py  pairwise_distances, bump
  * notebooks/tools/geostat.py       gaussian_fields, variogram_gauss  (stub for `mpl_tools.misc.nRowCol`)
  * notebooks/HistoryMatch.py        FunctionDefs ens_update0 (:578), ens_update0_loc (:774), IES (:906),
                                     ILES (:1007), vect (:413), perm_transf (:137), rms (:972) via `ast`

The RNG stream replays HistoryMatch.py exactly: rnd.seed(1) (:78) -> perm.Truth (:167) -> prod_noise (:261)
-> perm.Prior (:290) -> gg_setup (:600-603) -> hm_setup0 perturbs (:638).  The simulator (external package,
absent) is replaced where needed by a fixed linear observation operator; fixtures that would need it are not
produced.

Usage:  python oracle/make_golden.py
"""
import ast
import sys
import types
from pathlib import Path

import numpy as np
import numpy.random as rnd
import scipy.linalg as sla

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/notebooks")
OUT = ROOT / "tests" / "golden"


def import_reference():
    if not REF.exists():
        raise SystemExit("/root/reference not present: fixtures can only be regenerated in the build container")
    for name, attrs in (("adjustText", {"adjust_text": lambda *a, **k: None}),
                        ("mpl_tools", {}), ("mpl_tools.misc", {"nRowCol": lambda *a, **k: {}})):
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, str(REF))
    import tools.geostat as geostat
    import tools.localization as loc
    import tools.utils as utils

    class Dict(dict):  # stand-in for struct_tools.DotDict (attribute access only)
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    ns = dict(np=np, sla=sla, center=utils.center, utils=utils, Dict=Dict, nTime=40, sqrt=np.sqrt)
    tree = ast.parse((REF / "HistoryMatch.py").read_text())
    wanted = {"ens_update0", "ens_update0_loc", "IES", "ILES", "vect", "perm_transf", "rms"}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            exec(compile(ast.Module([node], []), "HistoryMatch.py", "exec"), ns)
    utils.progbar = lambda it, **k: it  # no tqdm bars
    return utils, loc, geostat, ns


def main():
    utils, loc, geostat, ns = import_reference()
    sys.path.insert(0, str(ROOT))
    from oracle.ressim import ResSim, default_wells

    OUT.mkdir(parents=True, exist_ok=True)
    ens_update0, ens_update0_loc, IES, ILES, vect = (ns[k] for k in ("ens_update0", "ens_update0_loc", "IES", "ILES", "vect"))
    nTime, nPrd, N = 40, 4, 40

    # ---- F1: seed-1 RNG replay (needs only the restated 20x20 mesh)
    model = default_wells(ResSim(20, 20, 2, 1))
    rnd.seed(1)
    geostat.randn = rnd.randn  # geostat imported `randn` from numpy.random: same global stream
    perm_truth = geostat.gaussian_fields(model.mesh, 1, r=0.8)
    # ---- F2: obs-error model (HistoryMatch.py:243-259, 639)
    corrs1well = np.exp(-np.arange(nTime) / 2)
    corrs1well[corrs1well < 1e-2] = 0
    R = np.kron(1e-2 * sla.toeplitz(corrs1well), np.eye(nPrd))
    R12 = sla.cholesky(R, lower=True)
    prod_noise = R12 @ rnd.randn(nTime * nPrd)
    perm_prior = geostat.gaussian_fields(model.mesh, N, r=0.8)
    d = 3
    gg_E = np.sqrt(4 / 3) * rnd.randn(400, d)
    gg_setup = dict(prior_ens=gg_E, obs=4 * np.ones(d), decorr=1 / np.sqrt(4) * np.eye(d),
                    perturbs=np.sqrt(4) * rnd.randn(*gg_E.shape))
    hm_perturbs = rnd.randn(N, nPrd * nTime) @ R12.T
    decorr = sla.inv(R12.T)
    np.savez_compressed(OUT / "f1_rng_replay.npz", perm_truth=perm_truth, prod_noise=prod_noise,
                        perm_prior=perm_prior, gg_E=gg_E, gg_perturbs=gg_setup["perturbs"], hm_perturbs=hm_perturbs)
    np.savez_compressed(OUT / "f2_obs_error.npz", R=R, R12=R12, decorr=decorr)

    # ---- F3: ens_update0
    gg_postr = ens_update0(**gg_setup, obs_ens=gg_E)
    H = np.random.RandomState(123).randn(400, nPrd * nTime) / 20  # fixed linear fake observation operator
    obs_ens = perm_prior @ H
    obs = (perm_truth @ H).ravel() + prod_noise
    hm = dict(obs_ens=obs_ens, obs=obs, perturbs=hm_perturbs, decorr=decorr)
    perm_es = ens_update0(perm_prior, **hm)
    es0 = ens_update0(obs_ens, **hm)  # data-space update, HistoryMatch.py:1156
    np.savez_compressed(OUT / "f3_ens_update0.npz", gg_postr=gg_postr, H=H, obs_ens=obs_ens, obs=obs,
                        perm_es=perm_es, es0=es0)

    # ---- F4: ens_update0_loc
    prod_inds = model.xy2ind(*model.prd_xy.T)
    xy_obs = np.tile(model.ind2xy(prod_inds), nTime)
    xy_prm = model.ind2xy(np.arange(model.Nxy))
    dists = loc.pairwise_distances(xy_prm.T, xy_obs.T)
    taper = loc.bump(dists / 1.2)
    gg_postr_loc = ens_update0_loc(**gg_setup, obs_ens=gg_E, taper=np.eye(d))
    les_ones = ens_update0_loc(perm_prior, **hm, taper=np.ones_like(dists))
    perm_les = ens_update0_loc(perm_prior, **hm, taper=taper)
    assert np.allclose(les_ones, perm_es)  # "Reproduces global analysis?" HistoryMatch.py:821-822
    np.savez_compressed(OUT / "f4_ens_update0_loc.npz", gg_postr_loc=gg_postr_loc, les_ones=les_ones,
                        perm_les=perm_les, distances_to_obs=dists, taper=taper, prod_inds=prod_inds)

    # ---- F5: helper known answers
    a = np.random.RandomState(5).randn(30, 7)
    b = np.random.RandomState(6).randn(30, 4)
    X, x = utils.center(a)
    Xr, _ = utils.center(a, rescale=True)
    sharp = np.array([0.01, 0.1, 1, 10, 100, 1000])  # HistoryMatch.py:687-690
    dd = np.linspace(-1, 1, 1001)
    bumps = np.array([loc.bump(dd, s) for s in sharp])
    A4 = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], float)
    np.savez_compressed(
        OUT / "f5_helpers.npz", a=a, b=b, center_X=X, center_x=x, center_Xr=Xr, cov=utils.cov(a, b),
        corr=utils.corr(a, b[:, 0]), bump_x=dd, bump_sharp=sharp, bumps=bumps, pd_A=A4,
        pd_AA=loc.pairwise_distances(A4), pd_1d=loc.pairwise_distances(np.arange(4)[:, None], [[2]]),
        pd_periodic=loc.pairwise_distances(np.arange(4)[:, None], domain=(4,)),
        vg=geostat.variogram_gauss(np.array([0, 1, 2]), 1, n=0.1, a=1))

    # ---- F6: iterative smoothers (for the "next" rows)
    ies, _ = IES(**gg_setup, obs_ens=lambda x: x)
    iles, _ = ILES(**gg_setup, obs_ens=lambda x: x, taper=np.eye(d))
    assert np.allclose(ies, gg_postr) and np.allclose(iles, gg_postr_loc)  # HistoryMatch.py:949-951, 1069-1071
    ies_lin, _ = IES(perm_prior, obs_ens=lambda x: x @ H, obs=obs, perturbs=hm_perturbs, decorr=decorr, xStep=0.4, iMax=3)
    np.savez_compressed(OUT / "f6_iterative.npz", ies_gg=ies, iles_gg=iles, ies_lin=ies_lin)
    print("gg posterior mean:", np.mean(gg_postr, 0), " prior mean/var:", perm_prior.mean(), perm_prior.var())
    print("first truth values:", perm_truth[0, :3])
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
