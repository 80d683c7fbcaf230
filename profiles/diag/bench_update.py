"""Analysis-step timing at BASELINE config 3 size (N=1000, M=128*128, n_obs=160): device time of the three update
phases (HIP events), MFMA rate of the two state-dimension contractions by the SURVEY 8d flop count 4*N*n_obs*M."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from historymatching_amd.update import UpdatePlan

N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
E = rng.randn(N, M); obs_ens = rng.rand(N, n_obs); obs = rng.rand(n_obs); perturbs = rng.randn(N, n_obs) * 0.1
decorr = np.eye(n_obs) * 3.0
res = {}
for dtype, use_mfma in ((32, 1), (32, 0), (64, 0)):
    p = UpdatePlan(N, N, M, n_obs, dtype=dtype)
    p.set_option("use_mfma", use_mfma)
    p.set_inputs(E, obs_ens, obs, perturbs, decorr)
    p.run_local()  # warm-up
    per_phase = []
    for ph in range(3):
        ts = []
        for _ in range(5):
            p.phase(ph); ts.append(p.sync()["ms_update"])
        per_phase.append(min(ts))
    tot = sum(per_phase)
    fused = min(p.run_local()["ms_update"] for _ in range(5))  # hm_upd_run: single call
    p.set_option("overlap", 1)
    fused2 = min(p.run_local()["ms_update"] for _ in range(5))  # ... with the small chain on a second stream
    p.set_option("overlap", 0)
    flops = 4.0 * N * n_obs * M
    res[f"dtype{dtype}_mfma{use_mfma}"] = {"ms_phase": per_phase, "ms_total": tot, "ms_fused_run": fused, "ms_fused_run_two_streams": fused2, "TFLOPs_fused_run": flops / fused / 1e9,
                                          "TFLOPs_min_flop_order": flops / tot / 1e9,
                                          "TFLOPs_contractions_only": flops / (per_phase[1] + per_phase[2]) / 1e9}
    p.close()
print(json.dumps(res, indent=1))
