"""Host algebra of the iterative ensemble smoother (update.ies_step; reference: IES, notebooks/HistoryMatch.py:927-942): the
Gram form (one LU solve + an n_obs x n_obs Cholesky factorisation) equals the reference's own evaluation
(pseudo-inverse of W, thin SVD of Y0) to rounding, for N > n_obs, N < n_obs and a W far from the identity; a singular W falls
back to the pseudo-inverse.  No device needed."""
import numpy as np
import pytest

from historymatching_amd.update import ies_step


@pytest.mark.parametrize("N,n_obs,spread", [(40, 160, 0.1), (400, 3, 0.1), (200, 160, 1.0)])
def test_gram_form_equals_pinv_svd_form(N, n_obs, spread):
    rng = np.random.RandomState(N + n_obs)
    W = np.eye(N) + spread * rng.randn(N, N) / np.sqrt(N)
    Eo, innov = rng.randn(N, n_obs), rng.randn(N, n_obs)
    a, b = ies_step(W, Eo, innov, "gram"), ies_step(W, Eo, innov, "svd")
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())


def test_singular_weights_fall_back_to_the_pseudo_inverse():
    rng = np.random.RandomState(3)
    N, n_obs = 30, 8
    W = np.eye(N)
    W[:, 5] = W[:, 4]  # rank N - 1
    Eo, innov = rng.randn(N, n_obs), rng.randn(N, n_obs)
    assert np.array_equal(ies_step(W, Eo, innov, "gram"), ies_step(W, Eo, innov, "svd"))
