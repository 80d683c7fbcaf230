"""k_sat128e (edge exchange, default) against k_sat128 (full fw image, sat_variant 5) at BASELINE config 2 size:
launch averages over a whole 40-step run and bitwise comparison of the final state and the producer series."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
out = {}
for v in (0, 5, 0):
    plan = ForwardPlan(model, N, bench.DT, nT, keep_history=False, device=0)
    plan.set_variant(0, v)
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    st = plan.sync()
    print(f"saturation variant {v}: {st['ms_saturation'] / st['n_saturation_launches']:.2f} ms/launch "
          f"(pressure {st['ms_pressure'] / st['n_pressure_launches']:.2f})", flush=True)
    out[v] = (plan.get_field("S").copy(), plan.outputs(want_wsats=False)[1].copy(), plan.get_field("nts").copy())
    plan.close()
for name, a, b in zip(("S", "prods", "nts"), out[0], out[5]):
    print(name, "bit-identical" if np.array_equal(a, b) else f"DIFFERENT max {np.abs(a - b).max()}")
