"""Whole-run parity at BASELINE config 2 shape: members of the synthetic prior advanced the full 40 steps on the GPU
(default kernels) and by the oracle; error against the oracle's own solver noise (COLAMD vs NATURAL ordering)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, oracle_sim_and_noise, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

n, N, steps = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 40
om, gm = make_models(n, n)
x = perms(n, n, N, seed=1)
plan = ForwardPlan(gm, N, 0.025, steps)
plan.set_inputs(x, transformed=False)
plan.run()
plan.sync()
w, p, status = plan.outputs()
plan.close()
assert not status.any()
for m in range(N):
    t0 = time.time()
    ref, noise = oracle_sim_and_noise(om, x[m], 0.025, steps)
    err = np.abs(w[m] - ref)
    print(f"member {m}: max|S_gpu - S_oracle| over 40 steps = {err.max():.3e} (at step {err.max(1).argmax()}), oracle COLAMD-vs-NATURAL "
          f"spread {noise:.3e}; producers max err {np.abs(p[m] - ref[1:, om.xy2ind(*om.prd_xy.T)]).max():.3e}; oracle time {time.time() - t0:.0f} s")
