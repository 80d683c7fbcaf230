"""End-to-end soft check through the product API on the GPU (the counterpart of oracle/run_reference_script.py, which runs the
reference's own script on the CPU oracle): a synthetic truth, noisy production data, a prior ensemble, then the ensemble
smoother, its localised form and the iterative smoother -- each must pull the ensemble's production towards the data and the
permeability towards the truth, as the reference's closing tables show for its own case (HistoryMatch.py:1187-1196)."""
import numpy as np
import pytest

from tests.helpers import perms, wells_4corners

pytestmark = pytest.mark.gpu

DT, NT, NX, N = 0.025, 40, 20, 200


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def test_history_matching_workflow_reduces_errors():
    from historymatching_amd.forward import make_forward_model
    from historymatching_amd.localization import taper_for_wells
    from historymatching_amd.obs import decorr as make_decorr, noisy_obs, obs_error_model, perturbations, vect
    from historymatching_amd.ressim import ResSim
    from historymatching_amd.update import ens_update0, ens_update0_loc, ies

    rng = np.random.RandomState(4)
    model = wells_4corners(ResSim(NX, NX, 2, 1))
    forward_model = make_forward_model(model, DT, NT)
    truth = perms(NX, NX, 1, seed=99)
    _, prod_truth = forward_model(truth)
    _, R12 = obs_error_model(NT, model.nPrd)
    obs = noisy_obs(vect(prod_truth[0]), R12, rng)
    prior = perms(NX, NX, N, seed=17)
    _, prod_prior = forward_model(prior)
    dec = make_decorr(R12)
    pert = perturbations(N, R12, rng)
    taper = taper_for_wells(model, model.xy2ind(*model.prd_xy.T), NT)

    posts = {
        "ES": ens_update0(prior, vect(prod_prior), obs, pert, dec),
        "LES": ens_update0_loc(prior, vect(prod_prior), obs, pert, dec, taper),
        "IES": ies(prior, lambda E: vect(forward_model(E)[1]), obs, pert, dec, xStep=0.4, iMax=4)[0],
    }
    err_prod_prior = rms(vect(prod_prior).mean(0) - vect(prod_truth[0]))
    err_perm_prior = rms(prior.mean(0) - truth[0])
    table = {}
    for name, E in posts.items():
        assert E.shape == prior.shape and np.isfinite(E).all()
        _, prod_post = forward_model(E)
        err_prod = rms(vect(prod_post).mean(0) - vect(prod_truth[0]))
        err_perm = rms(E.mean(0) - truth[0])
        spread = rms(E - E.mean(0))
        table[name] = (round(err_prod, 4), round(err_perm, 4), round(spread, 4))
    print("prior", round(err_prod_prior, 4), round(err_perm_prior, 4), round(rms(prior - prior.mean(0)), 4), table)
    for name, (err_prod, err_perm, spread) in table.items():
        assert err_prod < 0.7 * err_prod_prior, (name, err_prod, err_prod_prior)       # the data are matched better
        assert err_perm < err_perm_prior, (name, err_perm, err_perm_prior)              # and the field is closer to the truth
        assert spread < rms(prior - prior.mean(0)), name                                # with a narrower ensemble
