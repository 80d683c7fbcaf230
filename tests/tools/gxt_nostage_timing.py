import sys, numpy as np, scipy.linalg as sla
sys.path.insert(0, "/root/repo")
from historymatching_amd import _lib
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan
N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
R12 = obs_error_model(40, 4)[1]
p = UpdatePlan(N, N, M, n_obs, dtype=32)
p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
for dbg in (0, 1):
    p.set_option("gxt_debug", dbg)
    p.run_local()
    for _ in range(10):
        _lib.check(p.lib.hm_upd_run(p.h), "run")
    print("gxt_debug", dbg, "ms per step", p.sync()["ms_update"] / 10)
