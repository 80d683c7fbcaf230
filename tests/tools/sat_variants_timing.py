"""Saturation kernel variants at BASELINE config 2 size (N_e=1000, 128x128, fp64): launch averages."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = 1000
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
for v, name in ((0, "sat128 (register/LDS resident)"), (3, "tiled"), (2, "streaming"), (1, "generic")):
    plan = ForwardPlan(model, N, bench.DT, 4, keep_history=False, device=0)
    plan.set_variant(0, v)
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    st = plan.sync()
    print(f"saturation variant {v} ({name}): {st['ms_saturation'] / st['n_saturation_launches']:.1f} ms/launch")
    plan.close()
