"""Host-side observation helpers of the history-matching setup: tiny N x n_obs / n_obs x n_obs NumPy work that forms the
INPUTS of the update kernels.  Nothing here touches the GPU.

  vect              notebooks/HistoryMatch.py:413-421   flatten / unflatten the (time, well) axes of a production series
  obs_error_model   notebooks/HistoryMatch.py:243-259   R (exponentially time-correlated, independent between wells),
                                                        its lower Cholesky factor R12
  decorr            notebooks/HistoryMatch.py:638-639   inv(R12^T): right-multiplication whitens observation-space rows
  perturbations     notebooks/HistoryMatch.py:600-603   randn(N, n_obs) @ R12^T
  noisy_obs         notebooks/HistoryMatch.py:261-267   truth + R12 @ randn, clipped to [0, 1]
"""
import numpy as np
import scipy.linalg as sla


def vect(x, nTime=None, undo=False):
    """``(..., nTime, nPrd) -> (..., nTime*nPrd)``; ``undo=True`` restores the two axes (needs ``nTime``)."""
    x = np.asarray(x)
    if undo:
        if nTime is None:
            raise ValueError("vect(undo=True) needs nTime")
        *lead, ab = x.shape
        return x.reshape(list(lead) + [nTime, ab // nTime])
    *lead, a, b = x.shape
    return x.reshape(list(lead) + [a * b])


def obs_error_model(nTime, nPrd, corr_length=2.0, var=1e-2, cutoff=1e-2):
    """Observation-error covariance of the production series: ``var * exp(-|dt| / corr_length)`` in time (entries below
    ``cutoff`` dropped to keep R sparse-ish, as the reference does), independent between wells; returns ``(R, R12)``
    with ``R12`` the lower Cholesky factor."""
    c = np.exp(-np.arange(nTime) / corr_length)
    c[c < cutoff] = 0
    R = np.kron(var * sla.toeplitz(c), np.eye(nPrd))
    return R, sla.cholesky(R, lower=True)


def decorr(R12):
    """``inv(R12^T)``: ``obs_space_rows @ decorr`` has identity error covariance (what `ens_update0` expects)."""
    return sla.inv(np.asarray(R12).T)


def perturbations(N, R12, rng=None):
    """``N`` draws of the observation error, one per member (the perturbed-observation ensemble smoother)."""
    rng = np.random if rng is None else rng
    return rng.randn(N, len(R12)) @ np.asarray(R12).T


def noisy_obs(truth_series, R12, rng=None):
    """Synthetic observations: the truth's flattened production series plus one error draw, clipped to saturations."""
    rng = np.random if rng is None else rng
    return np.clip(np.asarray(truth_series) + np.asarray(R12) @ rng.randn(len(R12)), 0, 1)
