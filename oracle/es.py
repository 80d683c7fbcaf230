"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the ensemble-smoother update half.

PINNED: every function here is checked in tests/test_oracle_golden.py against fixtures captured from the
reference's own functions (imported / AST-extracted from /root/reference by oracle/make_golden.py, which also
re-asserts the reference's in-notebook identities HistoryMatch.py:821-822, 949-951, 1069-1071).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
Follows, line by line in meaning (not in text):
    center             notebooks/tools/utils.py:10-28
    cov, corr          notebooks/tools/utils.py:31-56
    ens_update0        notebooks/HistoryMatch.py:578-586
    ens_update0_loc    notebooks/HistoryMatch.py:774-797
    pairwise_distances notebooks/tools/localization.py:9-83
    bump               notebooks/tools/localization.py:86-92
    obs_error_model    notebooks/HistoryMatch.py:243-259, 639
    vect               notebooks/HistoryMatch.py:413-421
"""
import numpy as np
import scipy.linalg as sla


def center(E, axis=0, rescale=False):
    """Anomalies and mean of an ensemble along `axis` (utils.py:10-28)."""
    mean = np.mean(E, axis=axis, keepdims=True)
    anomalies = E - mean
    if rescale:
        n = E.shape[axis]
        anomalies *= np.sqrt(n / (n - 1))
    return anomalies, mean.squeeze()


def cov(a, b):
    """Sample cross-covariance of two ensembles with equal member count (utils.py:31-39)."""
    A = center(a)[0]
    B = center(b)[0]
    return A.T @ B / (len(B) - 1)


def corr(a, b):
    """Sample cross-correlation, clipped to +-999 (utils.py:42-56)."""
    c = cov(a, b)
    sa = np.std(a.T, axis=-1, ddof=1)
    sb = np.std(b, axis=0, ddof=1, keepdims=True)
    return (c / sa / sb).clip(-999, 999)


def ens_update0(prior_ens, obs_ens, obs, perturbs, decorr):
    """Stochastic ensemble-smoother update, decorrelated + transposed form (HistoryMatch.py:578-586).
    Same association as the reference: ((D pinv(C)) S^T) X."""
    N = len(prior_ens)
    X = center(prior_ens)[0]
    Y = center(obs_ens)[0]
    S = Y @ decorr
    D = (obs - obs_ens - perturbs) @ decorr
    C = S.T @ S + (N - 1) * np.eye(len(obs))
    return prior_ens + D @ sla.pinv(C) @ S.T @ X


def ens_update0_loc(prior_ens, obs_ens, obs, perturbs, decorr, taper, cutoff=1e-2):
    """Local analysis per state element with sqrt(taper) weights (HistoryMatch.py:774-797)."""
    N, M = prior_ens.shape
    X = center(prior_ens)[0]
    Y = center(obs_ens)[0]
    S = Y @ decorr
    D = (obs - obs_ens - perturbs) @ decorr
    out = np.empty_like(prior_ens, dtype=float)
    for i in range(M):
        ci = np.sqrt(taper[i])
        jj = ci > cutoff
        dE = 0
        if np.any(jj):
            Si = S[:, jj] * ci[jj]
            Di = D[:, jj] * ci[jj]
            Ci = Si.T @ Si + (N - 1) * np.eye(int(jj.sum()))
            dE = Di @ sla.pinv(Ci) @ Si.T @ X[:, i]
        out[:, i] = prior_ens[:, i] + dE
    return out


def pairwise_distances(A, B=None, domain=None):
    """Euclidean distances between point sets, optionally on a periodic box (localization.py:9-83)."""
    B = A if B is None else B
    A, B = np.atleast_2d(A), np.atleast_2d(B)
    assert A.shape[1] == B.shape[1]
    d = A[:, None] - B
    if domain:
        d = np.abs(d)
        d = np.minimum(d, np.reshape(domain, (1, 1, -1)) - d)
    return np.sqrt((d * d).sum(axis=-1)).reshape(len(A), len(B))


def bump(distances, sharpness=1):
    """exp(1 - 1/(1-x^2))**sharpness inside |x|<1, 0 outside (localization.py:86-92)."""
    distances = np.asarray(distances, dtype=float)
    inside = np.abs(distances) < 1
    x = distances[inside]
    out = np.zeros_like(distances)
    out[inside] = np.exp(1 - 1 / (1 - x * x)) ** sharpness
    return out


def obs_error_model(nTime, nPrd, length_tmp=2, var=1e-2):
    """R, its lower Cholesky factor, and decorr = inv(R12^T) (HistoryMatch.py:243-259, 639)."""
    c = np.exp(-np.arange(nTime) / length_tmp)
    c[c < 1e-2] = 0
    R = np.kron(var * sla.toeplitz(c), np.eye(nPrd))
    R12 = sla.cholesky(R, lower=True)
    return R, R12, sla.inv(R12.T)


def vect(x, nTime, undo=False):
    """Flatten/unflatten the (time, well) axes (HistoryMatch.py:413-421)."""
    if undo:
        *lead, ab = x.shape
        return x.reshape(list(lead) + [nTime, ab // nTime])
    *lead, a, b = x.shape
    return x.reshape(list(lead) + [a * b])


def es_mda(forward, prior_ens, obs, R12, n_iter, rng, nTime=None):
    """ES-MDA written as the loop SURVEY.md section 0.3/8f derives from the reference's pieces: `n_iter`
    passes of ens_update0 with the observation-error factor inflated by sqrt(alpha), alpha = n_iter
    (perturbs *= sqrt(alpha), decorr /= sqrt(alpha)), a fresh forward run and fresh perturbations per pass.
    `forward(E)` returns the observed ensemble (N, n_obs).  Not a reference function (the reference has no
    ES-MDA); oracle for the build's own driver."""
    E = np.array(prior_ens, dtype=float)
    alpha = float(n_iter)
    decorr = sla.inv(R12.T) / np.sqrt(alpha)
    for _ in range(n_iter):
        obs_ens = forward(E)
        perturbs = np.sqrt(alpha) * (rng.randn(len(E), len(obs)) @ R12.T)
        E = ens_update0(E, obs_ens, obs, perturbs, decorr)
    return E
