"""Saturation sweeps of the 128 x 128 fp64 path against each other: sat_variant 5 (fw image in LDS, sat128.hip), 1 (generic),
0 (fw in registers, scaled fluxes: sat128r.hip).  A whole forward run per variant with the same pressure kernel: S histories, producer series and
sub-step counts must be array_equal; then the launch average of each at N members.

    python tests/tools/sat_check.py [N=1000] [variants=5,0] [members=6] [steps=6]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "5,0").split(",")]
n = 128
_, gm = make_models(n, n)
M = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nT = int(sys.argv[4]) if len(sys.argv) > 4 else 6
x = perms(n, n, M, seed=3)
res = {}
for v in variants:
    plan = ForwardPlan(gm, M, 0.025, nT, keep_history=True, device=0)
    plan.set_variant(0, v)
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    w, pr, status = plan.outputs()
    res[v] = (w.copy(), pr.copy(), plan.get_field("nts").copy(), np.asarray(status).copy())
    print(f"sat_variant {v}: status {res[v][3].tolist()} nts[0] {res[v][2][0].tolist()}", flush=True)
    plan.close()
ref = variants[0]
ok = True
for v in variants[1:]:
    same = all(np.array_equal(a, b) for a, b in zip(res[ref], res[v]))
    ok &= same
    print(f"sat_variant {v} vs {ref}: {'array_equal' if same else 'DIFFERENT'}   max|dS| = {np.abs(res[ref][0] - res[v][0]).max():.3e}", flush=True)

if N > 0:
    xN = perms(n, n, N, seed=1)
    for v in variants:
        plan = ForwardPlan(gm, N, 0.025, 8, keep_history=False, device=0)
        plan.set_variant(0, v)
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        st = plan.sync()
        print(f"sat_variant {v}: saturation {st['ms_saturation'] / st['n_saturation_launches']:.3f} ms/launch, pressure "
              f"{st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch, {N} members, 8 steps", flush=True)
        plan.close()
sys.exit(0 if ok else 1)
