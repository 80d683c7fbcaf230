"""Forward model in dtype=32 mode (fp32 saturation, fp64 pressure) at BASELINE config 2/3 size: launch averages."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sat_variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # (as hm_fwd_set_variant)
model = bench.build_model(32, device=0)
plan = ForwardPlan(model, N, bench.DT, bench.NTIME, keep_history=False, device=0)
plan.set_variant(0, sat_variant)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
plan.set_inputs(x, None, transformed=False)
plan.run()
plan.sync()
plan.set_inputs(x, None, transformed=False)
plan.run()
st = plan.sync()
_, _, status = plan.outputs(want_wsats=False)
print(f"dtype=32, sat_variant {sat_variant}, {N} members: pressure {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms/launch, saturation "
      f"{st['ms_saturation'] / st['n_saturation_launches']:.2f} ms/launch, {N * bench.NTIME / (st['ms_total'] * 1e-3):.0f} ensemble-steps/s; "
      f"status ok {not status.any()}")
