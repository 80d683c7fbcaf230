"""Race check of the flag-synchronised chain kernels: the same analysis step many times, every output bit-identical to the first."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import scipy.linalg as sla
from historymatching_amd.obs import obs_error_model
from historymatching_amd.update import UpdatePlan

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for N, M, n_obs in ((1000, 128 * 128, 160), (257, 4096, 96), (4096, 8192, 160)):
    rng = np.random.RandomState(N)
    R12 = obs_error_model(n_obs // 4, 4)[1]
    p = UpdatePlan(N, N, M, n_obs, dtype=32)
    p.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
    p.run_local()
    ref = p.output().copy()
    bad = 0
    for r in range(reps):
        p.run_local()
        if not np.array_equal(p.output(), ref):
            bad += 1
    print(f"N={N} M={M} n_obs={n_obs}: {reps} repeats, {bad} differ from the first", flush=True)
    p.close()
