"""Wall time of the drop-in analysis call with host arrays in and out -- ens_update0(prior_ens, obs_ens, obs, perturbs, decorr)
(HistoryMatch.py:578-586) at BASELINE config 3's shape: 131 MB (fp64) of ensemble each way around 0.16 ms of device work."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.update import ens_update0  # noqa: E402

N, M, n_obs = 1000, 16384, 160
rng = np.random.RandomState(3)
E = rng.randn(N, M)
obs_ens = rng.rand(N, n_obs)
obs = rng.rand(n_obs)
R12 = 0.1 * np.eye(n_obs)
perturbs = rng.randn(N, n_obs) @ R12.T
decorr = np.linalg.inv(R12.T)
for dtype in (64, 32):
    ft = np.float64 if dtype == 64 else np.float32
    args = [np.ascontiguousarray(v, dtype=ft) for v in (E, obs_ens, obs, perturbs, decorr)]  # (fp64 arrays to an fp32 plan are converted on the host first)
    ens_update0(*args, dtype=dtype)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        out = ens_update0(*args, dtype=dtype)
        best = min(best, time.perf_counter() - t0)
    print(f"ens_update0 fp{dtype}, N={N}, M={M}, n_obs={n_obs}: best of 5 {1e3 * best:.1f} ms per call (host arrays in and out, "
          f"{E.nbytes * (dtype // 8) / 8 / 1e6:.0f} MB each way)", flush=True)
