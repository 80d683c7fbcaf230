"""Taper construction for the localised update: host NumPy, as in the reference (these are *inputs* of
``ens_update0_loc``; cheap, outside the device hot path).

Reference: ``pairwise_distances`` notebooks/tools/localization.py:9-83, ``bump`` :86-92, wiring
notebooks/HistoryMatch.py:700-717, 863.
"""

import numpy as np


def pairwise_distances(A, B=None, domain=None):
    """Euclidean distance between every point of ``A (nA, nDim)`` and of ``B (nB, nDim)``; ``domain`` makes the
    box periodic.  A 1-D input is a single point (localization.py:58-60)."""
    A = np.atleast_2d(A)
    B = A if B is None else np.atleast_2d(B)
    if A.shape[1] != B.shape[1]:
        raise AssertionError("The last axis of A and B must have equal length.")
    diff = A[:, None, :] - B[None, :, :]
    if domain:
        diff = np.abs(diff)
        diff = np.minimum(diff, np.reshape(domain, (1, 1, -1)) - diff)
    return np.sqrt(np.sum(diff * diff, axis=-1))


def bump(distances, sharpness=1):
    """Compactly supported taper ``exp(1 - 1/(1 - d^2))**sharpness`` for ``|d| < 1``, else 0."""
    d = np.asarray(distances, dtype=float)
    coeffs = np.zeros_like(d)
    m = np.abs(d) < 1
    dm = d[m]
    coeffs[m] = np.exp(1 - 1 / (1 - dm * dm)) ** sharpness
    return coeffs


def taper_for_wells(model, prod_inds, nTime, radius=1.2, sharpness=1):
    """``bump(distances_to_obs / radius)`` with the producer locations repeated per time (HistoryMatch.py:700-717,
    863) -> ``(Nxy, nPrd*nTime)``."""
    xy_obs = np.tile(model.ind2xy(prod_inds), nTime)
    xy_prm = model.ind2xy(np.arange(model.Nxy))
    return bump(pairwise_distances(xy_prm.T, xy_obs.T) / radius, sharpness)
