"""BASELINE config 1's forward pass (N_e = 100, 20 x 20, 40 steps) through the one-launch kernel of small grids with 64 / 128 / 256 threads per
member (hm_fwd_set_debug "small_wv"): device time per pass, results compared bit for bit.     python profiles/diag/config1_wv.py [N=100]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ref = None
for dtype in (64, 32):
    _, gm = make_models(20, 20, dtype=dtype)
    x = perms(20, 20, N, seed=1)
    ref = None
    for wv in (256, 128, 64, 0):
        plan = ForwardPlan(gm, N, 0.025, 40, keep_history=True)
        plan.set_debug("small_wv", wv)
        best = 1e9
        for _ in range(5):
            plan.set_inputs(x, None, transformed=False)
            plan.sync()
            t0 = time.perf_counter()
            plan.run()
            st = plan.sync()
            best = min(best, time.perf_counter() - t0)
        w, p, status = plan.outputs()
        plan.close()
        if ref is None:
            ref = (w, p)
        print(f"dtype {dtype}  small_wv {wv:3d}: {1e3 * best:6.2f} ms per pass (device {st['ms_total']:.2f} ms), status ok {not status.any()}, "
              f"identical to 256 threads: {np.array_equal(w, ref[0]) and np.array_equal(p, ref[1])}", flush=True)
