"""Analysis step at config 3's shape, batches of 10 back-to-back steps right after plan set-up: how long until the times settle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import scipy.linalg as sla
from historymatching_amd import _lib
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan

N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
R12 = obs_error_model(40, 4)[1]
p = UpdatePlan(N, N, M, n_obs, dtype=32)
p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
p.run_local()
out = []
t0 = time.perf_counter()
for b in range(80):
    for _ in range(10):
        _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
    out.append((time.perf_counter() - t0, p.sync()["ms_update"] / 10))
print(" ".join(f"{1e3 * t:.0f}ms:{1e3 * v:.0f}" for t, v in out[::4]))
