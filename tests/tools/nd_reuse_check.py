"""Nested-dissection pressure solve with and without the reuse of dry fronts across time steps (press_variant 12 / 14): whole runs must be
array_equal (saturation histories, producer series, sub-step counts); then the launch average of the pressure step over a whole run of
N members for both.
    python tests/tools/nd_reuse_check.py [N=1000] [members=8] [steps=40] [grid=128]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nT = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n = int(sys.argv[4]) if len(sys.argv) > 4 else 128
_, gm = make_models(n, n)
x = perms(n, n, M, seed=3)
res = {}
for v in (14, 12):
    plan = ForwardPlan(gm, M, 0.025, nT, keep_history=True, device=0)
    plan.set_variant(v, 0)
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    w, pr, status = plan.outputs()
    res[v] = (w.copy(), pr.copy(), plan.get_field("nts").copy(), plan.get_field("P").copy())
    # a second run on the same plan from fresh inputs: the cache must not leak from the end of the first run into the start of the second
    plan.set_inputs(x[::-1].copy(), None, transformed=False)
    plan.run()
    w2, pr2, _ = plan.outputs()
    res[(v, 2)] = (w2.copy(), pr2.copy())
    print(f"press_variant {v}: status {np.asarray(status).tolist()}", flush=True)
    plan.close()
ok = all(np.array_equal(a, b) for a, b in zip(res[14], res[12])) and all(np.array_equal(a, b) for a, b in zip(res[(14, 2)], res[(12, 2)]))
print(f"reuse (12) vs none (14): {'array_equal' if ok else 'DIFFERENT'} over {M} members x {nT} steps, twice; max|dS| = {np.abs(res[14][0] - res[12][0]).max():.3e}", flush=True)
if N > 0:
    xN = perms(n, n, N, seed=1)
    for v in (14, 12):
        plan = ForwardPlan(gm, N, 0.025, nT, keep_history=False, device=0)
        plan.set_variant(v, 0)
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        plan.sync()
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        st = plan.sync()
        print(f"press_variant {v}: pressure {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch averaged over {nT} steps, saturation "
              f"{st['ms_saturation'] / st['n_saturation_launches']:.3f} ms/launch, {N} members", flush=True)
        plan.close()
sys.exit(0 if ok else 1)
