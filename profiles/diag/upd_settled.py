"""Settled time of the analysis step at config 3's shape: 60 batches of 10 back-to-back steps, median of the last 30 (for A/B runs
with HM_AMD_LIB=...)."""
import os, sys
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
for a in sys.argv[1:]:
    k, v = a.split("=")
    p.set_option(k, int(v))
p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
p.run_local()
out = []
for b in range(60):
    for _ in range(10):
        _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
    out.append(p.sync()["ms_update"] / 10)
print(f"{os.environ.get('HM_AMD_LIB', 'default lib')}: settled {1e3 * np.median(out[30:]):.1f} us (first batch {1e3 * out[0]:.1f})")
