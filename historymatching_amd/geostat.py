"""Synthetic prior (pre-permeability) fields for grids the reference's dense sampler cannot reach.

The reference's ``geostat.gaussian_fields`` (notebooks/tools/geostat.py:86-99) builds the dense
``Nxy x Nxy`` Gaussian-variogram covariance and Cholesky-factors it: O(Nxy^3), ~15 min and >6 GB at 128^2,
infeasible beyond.  That covariance is exactly separable, ``Cov = Cx (x) Cy``, so
the same law is sampled here per axis.  Statistically the same prior, NOT bit-identical to the dense
sampler (whose output depends on LAPACK rounding of a numerically singular matrix).  Host NumPy: this is an
input generator, not part of the hot path.
"""

import numpy as np
import scipy.linalg as sla


def variogram_gauss(xx, r, n=0, a=1 / 3):
    """Gaussian variogram, same parametrisation as notebooks/tools/geostat.py:10-30."""
    xx = np.asarray(xx, dtype=float)
    gamma = (1 - n) * (1 - np.exp(-(xx**2) / r**2 / a))
    return np.where(xx != 0, gamma + n, gamma)


def gaussian_fields_kron(Nx, Ny, Lx, Ly, N=1, r=0.8, seed=None, rng=None):
    """``(N, Nx*Ny)`` zero-mean unit-variance Gaussian fields on the cell centres of an ``Nx x Ny`` grid."""
    if rng is None:
        rng = np.random.RandomState(seed)
    xc = (np.arange(Nx) + 0.5) * Lx / Nx
    yc = (np.arange(Ny) + 0.5) * Ly / Ny
    Cx = 1 - variogram_gauss(np.abs(xc[:, None] - xc), r)
    Cy = 1 - variogram_gauss(np.abs(yc[:, None] - yc), r)
    Ux = sla.cholesky(Cx + 1e-10 * np.eye(Nx))  # same nugget as geostat.py:97
    Uy = sla.cholesky(Cy + 1e-10 * np.eye(Ny))
    Z = rng.randn(N, Nx, Ny)
    return np.einsum("ki,nkl,lj->nij", Ux, Z, Uy, optimize=True).reshape(N, Nx * Ny)
