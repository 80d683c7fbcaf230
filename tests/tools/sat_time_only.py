"""Launch average of the default fp64 saturation sweep at config 2 size (for timing experiments with HM_AMD_LIB=...)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 40
v = int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
plan = ForwardPlan(model, N, bench.DT, nT, keep_history=False, device=0)
plan.set_variant(0, v)
plan.set_inputs(x, None, transformed=False)
plan.run()
st = plan.sync()
print(f"saturation variant {v}: {st['ms_saturation'] / st['n_saturation_launches']:.2f} ms/launch", flush=True)
