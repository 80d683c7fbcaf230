"""Pressure time per member of the larger grids' nested dissection against the member-block size (hm_fwd_set_debug "nd_cap"):  python tests/tools/ndl_cap_timing.py N cap"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
N, cap = int(sys.argv[1]), int(sys.argv[2])
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
n = 256
_, gm = make_models(n, n)
x = perms(n, n, N, seed=1)
plan = ForwardPlan(gm, N, 0.025, 3, keep_history=False, device=0)
plan.set_debug("nd_cap", cap)
plan.set_inputs(x, None, transformed=False)
plan.run(); plan.sync()
plan.set_inputs(x, None, transformed=False)
plan.run()
st = plan.sync()
print(f"N {N} cap {cap}: pressure {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms/launch = {st['ms_pressure'] / st['n_pressure_launches'] / N * 512:.2f} ms per 512 members; saturation {st['ms_saturation'] / st['n_saturation_launches'] / N * 512:.2f} ms per 512", flush=True)
plan.close()
