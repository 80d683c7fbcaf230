"""TEST INFRASTRUCTURE ONLY -- fixture F11: the reference's prior law on the default 20 x 20 grid.

Runs the REAL notebooks/tools/geostat.py (imported from /root/reference with a stand-in for the plotting-only `mpl_tools.misc.nRowCol`):
`vectorize`, `dist_euclid`, `variogram_gauss` give the dense covariance `Cov = 1 - variogram_gauss(dists, r=0.8)` of the 400 cell centres
exactly as `gaussian_fields` forms it (geostat.py:86-99, called at HistoryMatch.py:152-168, 289-290), and `gaussian_fields` itself draws
the seed-1 truth field of the script.  Stored: 12 rows of Cov (cells spread over the grid; every row is a function of distances, so these
pin the law), its diagonal, and the 2000-sample covariance of the reference sampler on those cells (what "statistically the same
prior" means at this sample size).  tests/test_oracle_golden.py checks oracle/geostat.py against it; tests/test_update_gpu.py checks the
device sampler hm_sample_kron against it.

Usage:  python oracle/make_golden_prior.py   (/root/reference must exist)"""
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/notebooks")
if not REF.exists():
    raise SystemExit("/root/reference not present: this fixture can only be regenerated in the build container")
for name, attrs in (("mpl_tools", {}), ("mpl_tools.misc", {"nRowCol": lambda *a, **k: {}})):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules.setdefault(name, mod)
sys.path.insert(0, str(REF))
import tools.geostat as geostat  # noqa: E402

Nx = Ny = 20
Lx, Ly, r = 2.0, 1.0, 0.8
xc, yc = (np.arange(Nx) + 0.5) * Lx / Nx, (np.arange(Ny) + 0.5) * Ly / Ny
mesh = np.meshgrid(xc, yc, indexing="ij")                     # model.mesh
dists = geostat.dist_euclid(geostat.vectorize(*mesh))
Cov = 1 - geostat.variogram_gauss(dists, r)
cells = np.array([0, 19, 21, 105, 190, 209, 210, 250, 333, 380, 398, 399])
geostat.randn = np.random.RandomState(7).randn              # the module-level name gaussian_fields draws from
fields = geostat.gaussian_fields(mesh, 2000, r=r)
sample_cov = (fields - fields.mean(0)).T @ (fields - fields.mean(0))[:, cells] / (len(fields) - 1)
np.savez_compressed(ROOT / "tests" / "golden" / "f11_prior_law.npz", Nx=Nx, Ny=Ny, Lx=Lx, Ly=Ly, r=r, cells=cells, cov_rows=Cov[cells],
                    cov_diag=np.diag(Cov).copy(), ref_sample_cov_2000=sample_cov.T, ref_sample_fields_first3=fields[:3])
print("Cov rows", Cov[cells].shape, "max |sample cov - Cov| over the stored rows at 2000 samples:", np.abs(sample_cov.T - Cov[cells]).max())
