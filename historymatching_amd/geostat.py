"""Synthetic prior (pre-permeability) fields for grids the reference's dense sampler cannot reach.

The reference's ``geostat.gaussian_fields`` (notebooks/tools/geostat.py:86-99) builds the dense
``Nxy x Nxy`` Gaussian-variogram covariance and Cholesky-factors it: O(Nxy^3), ~15 min and >6 GB at 128^2,
infeasible beyond.  That covariance is exactly separable, ``Cov = Cx (x) Cy``, so
the same law is sampled here per axis.  Statistically the same prior, NOT bit-identical to the dense
sampler (whose output depends on LAPACK rounding of a numerically singular matrix).  ``gaussian_fields_kron`` is host
NumPy; ``gaussian_fields_kron_device`` draws the same normals on the host (seeds replay) and does the two contractions
on the GPU's fp64 matrix cores (``hm_sample_kron``), optionally leaving the fields in HBM for a forward plan.
"""

import numpy as np
import scipy.linalg as sla


def variogram_gauss(xx, r, n=0, a=1 / 3):
    """Gaussian variogram, same parametrisation as notebooks/tools/geostat.py:10-30."""
    xx = np.asarray(xx, dtype=float)
    gamma = (1 - n) * (1 - np.exp(-(xx**2) / r**2 / a))
    return np.where(xx != 0, gamma + n, gamma)


def _axis_factors(Nx, Ny, Lx, Ly, r):
    xc = (np.arange(Nx) + 0.5) * Lx / Nx
    yc = (np.arange(Ny) + 0.5) * Ly / Ny
    Cx = 1 - variogram_gauss(np.abs(xc[:, None] - xc), r)
    Cy = 1 - variogram_gauss(np.abs(yc[:, None] - yc), r)
    Ux = sla.cholesky(Cx + 1e-10 * np.eye(Nx))  # same nugget as geostat.py:97
    Uy = sla.cholesky(Cy + 1e-10 * np.eye(Ny))
    return Ux, Uy


def gaussian_fields_kron(Nx, Ny, Lx, Ly, N=1, r=0.8, seed=None, rng=None):
    """``(N, Nx*Ny)`` zero-mean unit-variance Gaussian fields on the cell centres of an ``Nx x Ny`` grid."""
    if rng is None:
        rng = np.random.RandomState(seed)
    Ux, Uy = _axis_factors(Nx, Ny, Lx, Ly, r)
    Z = rng.randn(N, Nx, Ny)
    return np.einsum("ki,nkl,lj->nij", Ux, Z, Uy, optimize=True).reshape(N, Nx * Ny)


def gaussian_fields_kron_device(Nx, Ny, Lx, Ly, N=1, r=0.8, seed=None, rng=None, device=None, out_ptr=None, fetch=True):
    """The same fields (same normals for the same seed; the contractions are summed in a different order, so equal to
    ``gaussian_fields_kron`` within rounding, not bitwise) with the two contractions on the GPU.  ``out_ptr``: device address
    of an ``N*Nx*Ny`` fp64 buffer of this context to fill as well (``ForwardPlan.device_ptr``-style chaining);
    ``fetch=False`` skips the host copy and returns None."""
    import ctypes as C

    from . import _lib

    if rng is None:
        rng = np.random.RandomState(seed)
    Ux, Uy = (np.ascontiguousarray(a) for a in _axis_factors(Nx, Ny, Lx, Ly, r))
    Z = np.ascontiguousarray(rng.randn(N, Nx, Ny))
    ctx = _lib.Context.get(device)
    out = np.empty((N, Nx * Ny)) if fetch else None
    if out is None and out_ptr is None:
        raise ValueError("nothing to produce: fetch=False and no out_ptr")
    _lib.check(ctx.lib.hm_sample_kron(ctx.handle, int(N), int(Nx), int(Ny), _lib.ptr(Ux), _lib.ptr(Uy), _lib.ptr(Z),
                                      None if out is None else _lib.ptr(out), None if out_ptr is None else C.c_void_p(out_ptr)),
               "hm_sample_kron")
    return out
