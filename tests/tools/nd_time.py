"""Launch averages of one pressure variant over a whole forward run of N members (for rocprofv3 --kernel-trace --stats): the nested
dissection (12; 14 = without the reuse of dry fronts) does less work while the grid is still dry ahead of the front, so a single
solve on the initial state says nothing about a run.
    python tests/tools/nd_time.py [variant=12] [N=1000] [runs=1]"""
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
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_, gm = make_models(128, 128)
x = perms(128, 128, N, seed=1)
plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False, device=0)
plan.set_variant(v, 0)
plan.set_inputs(x, None, transformed=False)
plan.run()  # warm-up run: lazily allocated buffers
plan.sync()
t0 = time.perf_counter()
for _ in range(runs):
    plan.set_inputs(x, None, transformed=False)
    plan.run()
st = plan.sync()
print(f"variant {v}: {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch (events) averaged over {runs} run(s) of 40 time steps, "
      f"{(time.perf_counter() - t0) / runs:.3f} s wall per run, {N} members", flush=True)
plan.close()
