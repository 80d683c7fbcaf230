"""Does a forward run earlier in the process slow the analysis step down?  (scratch memory on the queue / clocks)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import scipy.linalg as sla
import bench
from historymatching_amd import _lib
from historymatching_amd.forward import ForwardPlan
from historymatching_amd.geostat import gaussian_fields_kron
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan

N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
R12 = obs_error_model(40, 4)[1]
p = UpdatePlan(N, N, M, n_obs, dtype=32)
p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
p.run_local()

def series(tag, nb=24):
    out = []
    for b in range(nb):
        for _ in range(10):
            _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
        out.append(p.sync()["ms_update"] / 10)
    print(tag, " ".join(f"{1e3 * v:.0f}" for v in out[::3]), flush=True)

series("before any forward run:")
members = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = bench.build_model(64, device=0)
fp = ForwardPlan(model, members, bench.DT, 4, keep_history=False, device=0)
fp.set_inputs(gaussian_fields_kron(128, 128, 2, 1, members, r=0.8, seed=1), None, transformed=False)
fp.run(); fp.sync()
series(f"after a 4-step forward run of {members} members (plan alive):")
fp.close()
series("after closing the forward plan:")
time.sleep(2.0)
series("2 s later:")
