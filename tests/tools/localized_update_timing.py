"""Localised analysis (ens_update0_loc) timing at config-3/5-like sizes: taper = bump(dist / 1.2) between all cells of a
128x128 grid and the 4 producers x 40 times (HistoryMatch.py:700-717, 863)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.localization import bump, pairwise_distances  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402
from historymatching_amd.update import UpdatePlan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, nTime = 1000, 40
model = ResSim(n, n, 2, 1)
near01 = np.array([0.12, 0.87])
model.prd_xy = [[x, y] for y in model.Ly * near01 for x in model.Lx * near01]
prod_inds = model.xy2ind(*model.prd_xy.T)
xy_obs = np.tile(model.ind2xy(prod_inds), nTime)
xy_prm = model.ind2xy(np.arange(model.Nxy))
taper = bump(pairwise_distances(xy_prm.T, xy_obs.T) / 1.2)
M, n_obs = taper.shape
rng = np.random.RandomState(0)
for dtype in (32, 64):
    p = UpdatePlan(N, N, M, n_obs, dtype=dtype, localized=True)
    p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), 0.1 * rng.randn(N, n_obs), 3.0 * np.eye(n_obs), taper=taper)
    p.run_local()
    ts = []
    for ph in range(3):
        t = []
        for _ in range(3):
            p.phase(ph)
            t.append(p.sync()["ms_update"])
        ts.append(min(t))
    nloc = (np.sqrt(taper) > 1e-2).sum(1)
    print(f"dtype {dtype}: M={M}, n_obs={n_obs}, n_loc mean {nloc.mean():.0f} max {nloc.max()}: phases {['%.3f' % x for x in ts]} ms, total {sum(ts):.3f} ms")
    p.close()
