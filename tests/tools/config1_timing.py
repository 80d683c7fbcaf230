"""BASELINE config 1's forward run on the GPU (N_e = 100, 20 x 20, 40 steps): wall per forward pass through the drop-in and through a
device-resident plan, and the per-launch device times.   python tests/tools/config1_timing.py [N=100] [repeats=20]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.forward import ForwardPlan, make_forward_model  # noqa: E402
from tests.helpers import make_models, perms  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
generic = len(sys.argv) > 3 and sys.argv[3] == "generic"   # the per-step generic kernels (pressure / saturation variant 1) instead of the single launch
_, gm = make_models(20, 20)
x = perms(20, 20, N, seed=1)
fm = make_forward_model(gm, 0.025, 40)
fm(x)
t0 = time.perf_counter()
for _ in range(reps):
    w, p = fm(x)
t_host = (time.perf_counter() - t0) / reps
plan = ForwardPlan(gm, N, 0.025, 40, keep_history=True)
if generic:
    plan.set_variant(1, 1)
plan.set_inputs(x, None, transformed=False)
plan.run()
plan.sync()
t0 = time.perf_counter()
for _ in range(reps):
    plan.set_inputs(x, None, transformed=False)
    plan.run()
    st = plan.sync()
t_plan = (time.perf_counter() - t0) / reps
print(f"config 1 forward (N = {N}, 20 x 20, 40 steps{', generic per-step kernels' if generic else ''}): drop-in call {t_host * 1e3:.2f} ms, device-resident plan {t_plan * 1e3:.2f} ms per pass; "
      f"device: total {st['ms_total']:.2f} ms, pressure {st['ms_pressure'] / max(1, st['n_pressure_launches']) * 1e3:.1f} us / launch, "
      f"saturation {st['ms_saturation'] / max(1, st['n_saturation_launches']) * 1e3:.1f} us / launch")
plan.close()
