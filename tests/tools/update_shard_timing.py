"""Per-GPU cost of the row-sharded analysis step at the shapes of BASELINE configs 4 and 5 (one rank's share; the two
all-reduces in between are reported by size -- a single box has no peers):
  config 4: N_e=4096 over 8 GPUs -> 512 local members, M=256*256, n_obs=160, global update, fp32 state;
  config 5: N_e=1000 over 8 GPUs -> 125 local members, M=512*512, localised (taper=bump(dist/1.2)), fp32."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.localization import bump, pairwise_distances  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402
from historymatching_amd.update import UpdatePlan  # noqa: E402

rng = np.random.RandomState(0)
for name, Ntot, Nloc, n, localized in (("config 4 shard", 4096, 512, 256, False), ("config 5 shard", 1000, 125, 512, True)):
    M, n_obs = n * n, 160
    taper = None
    if localized:
        model = ResSim(n, n, 2, 1)
        near01 = np.array([0.12, 0.87])
        model.prd_xy = [[a, b] for b in model.Ly * near01 for a in model.Lx * near01]
        xy_obs = np.tile(model.ind2xy(model.xy2ind(*model.prd_xy.T)), 40)
        taper = bump(pairwise_distances(model.ind2xy(np.arange(M)).T, xy_obs.T) / 1.2).astype(np.float32)
    plan = UpdatePlan(Ntot, Nloc, M, n_obs, dtype=32, localized=localized)
    plan.set_inputs(rng.randn(Nloc, M).astype(np.float32), rng.rand(Nloc, n_obs), rng.rand(n_obs), 0.1 * rng.randn(Nloc, n_obs),
                    3.0 * np.eye(n_obs), taper=taper)
    for ph in range(3):
        plan.phase(ph)
    plan.sync()
    ts = []
    for ph in range(3):
        t = []
        for _ in range(3):
            plan.phase(ph)
            t.append(plan.sync()["ms_update"])
        ts.append(min(t))
    sizes = []
    for which in (0, 1, 2, 3):
        _, cnt, dt = plan.reduce_buffer(which)
        sizes.append(cnt * np.dtype(dt).itemsize)
    print(f"{name}: N_local={Nloc} of {Ntot}, M={M}, n_obs={n_obs}, localised={localized}, fp32: phases {['%.3f' % x for x in ts]} ms (sum {sum(ts):.3f} ms); "
          f"all-reduce after phase 0: {(sizes[0] + sizes[1]) / 1e6:.2f} MB, after phase 1: {(sizes[2] + sizes[3]) / 1e6:.1f} MB", flush=True)
    plan.close()
