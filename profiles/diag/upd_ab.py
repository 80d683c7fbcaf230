"""Back-to-back analysis step at config 3's shape for the values of an option given on the command line: option v1 v2 ... (5 rounds each, interleaved)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import scipy.linalg as sla
from historymatching_amd import _lib
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
R12 = obs_error_model(40, 4)[1]
args = (rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
plans = {}
for v in vals:
    p = UpdatePlan(N, N, M, n_obs, dtype=32)
    p.set_option(opt, v)
    p.set_inputs(*args)
    p.run_local()
    plans[v] = p
res = {v: [] for v in vals}
for r in range(5):
    for v in vals:
        p = plans[v]
        for _ in range(20):
            _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
        res[v].append(p.sync()["ms_update"] / 20)
for v in vals:
    print(f"{opt}={v}: " + " ".join(f"{t * 1e3:.1f}" for t in res[v]) + f"  us; median {np.median(res[v]) * 1e3:.1f} = {4.0 * N * n_obs * M / np.median(res[v]) / 1e9 / 157.3 * 100:.1f} % of the fp32 matrix peak")
