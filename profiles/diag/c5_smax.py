"""Config-5 shard (125 members, 512 x 512, dtype = 32 plans, 40 steps): how far the float32 saturation of a cell ends above 1, and how the
float32 and fp64 pressure fields of the first step compare between two builds:   HM_AMD_LIB=... python profiles/diag/c5_smax.py [N=125]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125
n, steps = 512, 40
_, gm = make_models(n, n, dtype=32)
x = perms(n, n, N, seed=8)
plan = ForwardPlan(gm, N, 0.025, steps, keep_history=False)
plan.set_inputs(x, transformed=False)
plan.run(0, 1)
plan.sync()
P1 = plan.get_field("P")
np.save(f"gpurun_out/c5_P1_{sys.argv[2] if len(sys.argv) > 2 else 'x'}.npy", P1[:4])
plan.run(1, steps - 1)
plan.sync()
S, prods, status = plan.outputs()
S = S.astype(float)
mx = S.max(1)
print("status any:", status.any(), " S max over members: max %.3e  median %.3e  members above 1 + 1e-4: %d of %d" % (mx.max() - 1, np.median(mx) - 1, (mx > 1 + 1e-4).sum(), N))
print("sorted excess (top 8):", np.sort(mx - 1)[-8:])
print("cells above 1 + 1e-5 per member (top 8):", np.sort((S > 1 + 1e-5).sum(1))[-8:])
