"""Launch average of one pressure variant at N members (for rocprofv3 --kernel-trace --stats):
    python tests/tools/nd_time.py [variant=12] [N=1000] [reps=10]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
_, gm = make_models(128, 128)
plan = ForwardPlan(gm, N, 0.025, 4, keep_history=False, device=0)
plan.set_variant(v, 0)
plan.set_inputs(perms(128, 128, N, seed=1), None, transformed=False)
for _ in range(3):
    plan.pressure_only(0)
plan.sync()
t0 = time.perf_counter()
for _ in range(reps):
    plan.pressure_only(0)
st = plan.sync()
print(f"variant {v}: {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch (events), {(time.perf_counter() - t0) / reps * 1e3:.3f} ms wall, {N} members", flush=True)
plan.close()
