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


def rectangular_partitioning(shape, steps, do_ind=True):
    """Rectangular batches ("local domains") of an N-D grid, as notebooks/tools/localization.py:95-145: along every axis the
    ``shape[d]`` indices are split into ``round(shape[d] / steps[d])`` nearly equal consecutive groups (``np.array_split``
    sizes: the first ``n % k`` groups get one more), a batch is the product of one group per axis, batches are ordered with
    the LAST axis varying fastest and so are the cells inside a batch.  Returns a list of flat C-order index arrays, or
    (``do_ind=False``) of per-axis coordinate lists."""
    if len(shape) != len(steps):
        raise AssertionError("shape and steps must have the same length")
    groups = []
    for n, step in zip(shape, steps):
        k = round(n / step)
        base, extra = divmod(int(n), k)
        sizes = [base + 1] * extra + [base] * (k - extra)
        edges = np.concatenate([[0], np.cumsum(sizes)])
        groups.append([np.arange(edges[i], edges[i + 1]) for i in range(k)])
    batches = []
    counts = [len(g) for g in groups]
    for flat in range(int(np.prod(counts))):
        which = np.unravel_index(flat, counts)                 # last axis fastest == itertools.product order
        axes = [groups[d][which[d]] for d in range(len(shape))]
        grids = np.meshgrid(*axes, indexing="ij")
        coords = [g.reshape(-1) for g in grids]
        batches.append(np.ravel_multi_index(coords, shape) if do_ind else coords)
    return batches
