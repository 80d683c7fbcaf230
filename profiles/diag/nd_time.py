"""Wall time of the nested-dissection pressure step (every front eliminated: press_variant 14), for A/B copies of the library:
     HM_AMD_LIB=build_ab/libhm_<name>.so python profiles/diag/nd_time.py [N=1000] [launches=10] [what=both|all|run]
("all": only the launches with every front eliminated; "run": only the 40-step runs with the dry-front reuse -- for a kernel trace of one of them)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
WHAT = sys.argv[3] if len(sys.argv) > 3 else "both"
_, gm = make_models(128, 128)
if WHAT == "run":
    L = 0
plan = ForwardPlan(gm, N, 0.025, 4, keep_history=False, device=0)
plan.set_variant(14, 0)
plan.set_inputs(perms(128, 128, N, seed=1), None, transformed=False)
for _ in range(3 if L else 0):
    plan.pressure_only(0)
plan.sync()
best = 1e9
for _ in range(3 if L else 0):
    t0 = time.perf_counter()
    for _ in range(L):
        plan.pressure_only(0)
    plan.sync()
    best = min(best, (time.perf_counter() - t0) / L)
if L:
    print(f"pressure step, every front: {1e3 * best:.3f} ms per launch of {N} members")
plan.close()
if WHAT == "all":
    sys.exit(0)
# the same over a run of 40 time steps with the default solver (dry fronts keep their results)
plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False, device=0)
plan.set_inputs(perms(128, 128, N, seed=1), None, transformed=False)
P = perms(128, 128, N, seed=1)
plan.run(0, 40)
plan.sync()
best = None
for _ in range(2):
    plan.set_inputs(P, None, transformed=False)
    plan.run(0, 40)
    st = plan.sync()
    v = st["ms_pressure"] / st["n_pressure_launches"]
    best = v if best is None else min(best, v)
print(f"pressure step over a run of 40 (dry fronts reused): {best:.3f} ms per launch (device time)")
