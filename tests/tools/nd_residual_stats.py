"""Distribution of the direct solver's a-posteriori residual max |div V - q| over an ensemble (the larger grids' nested dissection):
what the flux check of k_nd_flux (press_nd.hip) sees.   python tests/tools/nd_residual_stats.py [n=256 [N=1024 [seed=1000]]]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
om, gm = make_models(n, n)
q = om.source_field(0)[0]
x = gaussian_fields_kron(n, n, 2, 1, N, r=0.8, seed=seed).astype(np.float32).astype(np.float64)
plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False, device=0)
plan.set_inputs(x, None, transformed=False)
worst = np.zeros(N)
for k in range(40):
    plan.run(k, 1)
    st = plan.sync()
    if k % 6 == 0 or k == 39:
        Vx, Vy = plan.get_field("Vx"), plan.get_field("Vy")
        r = np.zeros(N)
        for m0 in range(0, N, 128):
            div = (Vx[m0:m0 + 128, 1:] - Vx[m0:m0 + 128, :-1]) + (Vy[m0:m0 + 128, :, 1:] - Vy[m0:m0 + 128, :, :-1])
            r[m0:m0 + 128] = np.abs(div.reshape(len(div), -1) - q).max(1)
        worst = np.maximum(worst, r)
        hist = [(r > t).sum() for t in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4)]
        print(f"step {k}: members with residual > 1e-10..1e-4: {hist}; max {r.max():.2e}; fallbacks so far {st['nd_fallbacks']}", flush=True)
_, _, status = plan.outputs(want_wsats=False)
print("status nonzero:", np.flatnonzero(status)[:20], "Kmax of the 5 worst:", [f"{(0.1 + np.exp(5 * x[m])).max():.2e}" for m in np.argsort(worst)[-5:]], "their residuals", np.sort(worst)[-5:])
plan.close()
