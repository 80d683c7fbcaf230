"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's prior sampler (notebooks/tools/geostat.py:10-30, 86-99):
dense Gaussian-variogram covariance of all cell centres, Cholesky factor of Cov + 1e-10 I, fields = randn(N, Nxy) @ C12.  O(Nxy^3):
the default 20 x 20 grid and small test grids only.  Pinned by tests/golden/f11_prior_law.npz (the reference's own Cov rows, captured by
oracle/make_golden_prior.py).  The product's scalable samplers (historymatching_amd/geostat.py: per-axis factors, hm_sample_kron on the
device) are checked against this law."""
import numpy as np
import scipy.linalg as sla


def variogram_gauss(xx, r, n=0, a=1 / 3):
    xx = np.asarray(xx, dtype=float)
    gamma = (1 - np.exp(-(xx**2) / r**2 / a)) * (1 - n)
    gamma[xx != 0] += n
    return gamma


def cell_centres(Nx, Ny, Lx, Ly):
    xc, yc = (np.arange(Nx) + 0.5) * Lx / Nx, (np.arange(Ny) + 0.5) * Ly / Ny
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    return np.stack([X.ravel(), Y.ravel()], 1)  # row k = centre of cell k = ix * Ny + iy


def covariance(Nx, Ny, Lx, Ly, r):
    pts = cell_centres(Nx, Ny, Lx, Ly)
    diff = pts[:, None, :] - pts
    return 1 - variogram_gauss(np.sqrt(np.sum(diff**2, axis=-1)), r)


def gaussian_fields(Nx, Ny, Lx, Ly, N=1, r=0.2, rng=None):
    rng = rng or np.random
    Cov = covariance(Nx, Ny, Lx, Ly, r)
    C12 = sla.cholesky(Cov + 1e-10 * np.eye(len(Cov)))
    return rng.randn(N, len(C12)) @ C12
