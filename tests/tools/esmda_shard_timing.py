"""Whole ES-MDA assimilation (4 passes) on one GPU at the per-GPU shapes of BASELINE configs 4 and 5, members resident in HBM
(dist.es_mda_sharded with one rank: the all-reduces of the real 8-rank run are not in these numbers):
  config 4 shard: 512 members, 256 x 256, global analysis, fp32 forward sweep + fp32 matrix-core update;
  config 5 shard: 125 members, 512 x 512, localised analysis (taper = bump(dist / 1.2)), fp32.
   python tests/tools/esmda_shard_timing.py [c4|c5]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import wells_4corners  # noqa: E402
from historymatching_amd.dist import es_mda_sharded  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402
from historymatching_amd.localization import taper_for_wells  # noqa: E402
from historymatching_amd.obs import obs_error_model  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

DT, NT = 0.025, 40
which = sys.argv[1:] or ["c4", "c5"]
for name, n, N, localized in (("c4", 256, 512, False), ("c5", 512, 125, True)):
    if name not in which:
        continue
    model = wells_4corners(ResSim(n, n, 2, 1, dtype=32))
    prior = gaussian_fields_kron(n, n, 2, 1, N, r=0.8, seed=1)
    _, R12 = obs_error_model(NT, model.nPrd)
    obs = np.clip(0.2 + 0.05 * np.random.RandomState(2).randn(NT * model.nPrd), 0, 1)
    taper = taper_for_wells(model, model.xy2ind(*model.prd_xy.T), NT).astype(np.float32) if localized else None
    st = {}
    t0 = time.perf_counter()
    post = es_mda_sharded(model, prior, obs, R12, DT, NT, n_iter=4, seed=3, dtype=32, taper=taper, stats=st)
    wall = time.perf_counter() - t0
    print(f"{name} shard: {N} members, {n}x{n}, {'localised' if localized else 'global'} ES-MDA, 4 passes, fp32: wall {wall:.1f} s, device forward "
          f"{st['ms_forward'] / 1e3:.1f} s + update {st['ms_update']:.1f} ms; {4 * N * NT / wall:.0f} ensemble-steps/s incl. updates; "
          f"posterior finite: {bool(np.isfinite(post).all())}, max |posterior - prior| = {np.abs(post - prior).max():.2f}", flush=True)
