"""The blocked device step of the iterative smoothers alone (for rocprofv3 --kernel-trace --stats): N members, B domains, n_obs observations.
    python tests/tools/ies_step_profile.py [N=1000] [B=1] [n_obs=160] [reps=3]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.update import IlesPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_obs = int(sys.argv[3]) if len(sys.argv) > 3 else 160
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.RandomState(0)
M = 64 * B
prior = rng.randn(N, M)
batches = [np.arange(64 * b, 64 * (b + 1)) for b in range(B)]
plan = IlesPlan(prior, batches, np.ones((B, n_obs)), cutoff=0.5)
S = rng.randn(N, n_obs)
S -= S.mean(0)
D = rng.randn(N, n_obs)
plan.step(S, D, 0.3)
t0 = time.perf_counter()
for _ in range(reps):
    plan.step(S, D, 0.3)
dt = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
E = plan.compose()
tc = time.perf_counter() - t0
print(f"N = {N}, {B} domain(s) of 64 elements, n_obs = {n_obs}: hm_iles_step {dt * 1e3:.2f} ms per call (host call incl. the copy of S, D in), "
      f"hm_iles_compose {tc * 1e3:.2f} ms; finite: {bool(np.isfinite(E).all())}")
plan.close()
