"""Analysis step (fused run, fp32 plans) at several shapes with the kernel variants behind `hm_upd_set_option`:
   apply_variant 1 (k_apply_lds) | 2 (k_apply_lds2) | 3 (k_apply_dma);  gxt_dma 0 (k_gxt_lds) | 1 (k_gxt_dma).
   python tests/tools/update_variants_timing.py"""
import sys
from pathlib import Path

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.obs import obs_error_model  # noqa: E402
from historymatching_amd.update import UpdatePlan  # noqa: E402

n_obs = 160
R12 = obs_error_model(40, 4)[1]
decorr = sla.inv(R12.T)
for N, M in ((1000, 128 * 128), (512, 256 * 256), (4096, 256 * 256), (1000, 512 * 512)):
    rng = np.random.RandomState(0)
    E = rng.randn(N, M).astype(np.float32)
    obs_ens = rng.rand(N, n_obs)
    p = UpdatePlan(N, N, M, n_obs, dtype=32)
    p.set_inputs(E, obs_ens, rng.rand(n_obs), rng.randn(N, n_obs) @ R12.T, decorr)
    flops = 4.0 * N * n_obs * M
    for av in (1, 2, 3):
        for dma in (0, 1):
            p.set_option("apply_variant", av)
            p.set_option("gxt_dma", dma)
            p.run_local()
            reps = 5
            for _ in range(reps):
                _lib.check(p.lib.hm_upd_run(p.h), "hm_upd_run")
            ms = p.sync()["ms_update"] / reps
            print(f"N={N:5d} M={M:7d}  apply_variant={av} gxt_dma={dma}: {ms:8.3f} ms  {flops / ms / 1e9 / 157.3:6.1%} of the fp32 matrix peak", flush=True)
    p.close()
