"""Launch average of the default pressure solve at config 2 size, on the initial saturation (for timing experiments with
HM_AMD_LIB=...; no saturation sweep runs, so a deliberately wrong solve cannot run away in sub-steps)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
plan = ForwardPlan(model, N, bench.DT, 4, keep_history=False, device=0)
plan.set_inputs(x, None, transformed=False)
for _ in range(5):
    plan.pressure_only(0)
plan.sync()
for _ in range(reps):
    plan.pressure_only(0)
st = plan.sync()
print(f"pressure: {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch over {st['n_pressure_launches']} launches, {N} members", flush=True)
