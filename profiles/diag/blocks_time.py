"""Config 2 (N_e = 1000, 128 x 128, fp64, 40 time steps) as member blocks on streams of their own: wall time per pass for a few
splits (BlockedForwardPlan).     python profiles/diag/blocks_time.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from historymatching_amd.forward import BlockedForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = 1000
DT_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = bench.build_model(DT_, device=0)
perms = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
ref = None
SPLITS = ([0, 1000], [0, 500, 1000], [0, 334, 667, 1000], [0, 250, 500, 750, 1000]) if DT_ == 32 else ([0, 1000], [0, 500, 1000], [0, 334, 667, 1000], [0, 256, 512, 1000], [0, 400, 700, 1000], [0, 250, 500, 750, 1000], [0, 200, 400, 600, 800, 1000], [0, 167, 334, 500, 667, 834, 1000])
if len(sys.argv) > 2:  # custom partitions: "0,256,512,1000;0,500,1000"
    SPLITS = [[int(v) for v in part.split(",")] for part in sys.argv[2].split(";")]
for bounds in SPLITS:
    plan = BlockedForwardPlan(model, N, bench.DT, bench.NTIME, keep_history=False, bounds=bounds)
    best = 1e9
    for rep in range(3):
        plan.set_inputs(perms, None, transformed=False)
        plan.sync()
        t0 = time.perf_counter()
        plan.run(0, bench.NTIME)
        plan.sync()
        best = min(best, time.perf_counter() - t0)
    p = plan.outputs(want_wsats=False)[1]
    if ref is None:
        ref = p
    print(f"blocks {bounds}: {1e3 * best:7.1f} ms per pass = {N * bench.NTIME / best / 1e3:6.2f} k ensemble-steps/s; producer series identical to one block: {np.array_equal(p, ref)}", flush=True)
    plan.close()
