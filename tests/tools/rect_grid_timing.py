"""Timing of a rectangular grid: python tests/tools/rect_grid_timing.py nx ny members steps [dtype=64] [embed=1] [force_square=0]
force_square: run the same model embedded by hand is not possible from here -- the third flag only switches the library's own choice off (embed 0)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import perms, wells_4corners  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

nx, ny, N, nTime = (int(a) for a in sys.argv[1:5])
dtype = int(sys.argv[5]) if len(sys.argv) > 5 else 64
embed = int(sys.argv[6]) if len(sys.argv) > 6 else 1
gm = wells_4corners(ResSim(nx, ny, 2, 1, dtype=dtype))
plan = ForwardPlan(gm, N, 0.025, nTime + 1, keep_history=False, device=0)
plan.set_debug("embed", embed)
plan.set_inputs(perms(nx, ny, N, seed=3), None, transformed=False)
plan.run(0, 1)
plan.sync()
t0 = time.perf_counter()
plan.run(1, nTime)
st = plan.sync()
wall = time.perf_counter() - t0
_, _, status = plan.outputs(want_wsats=False)
print(f"{nx}x{ny}, {N} members, {nTime} steps after one warm-up step, dtype {dtype}, embed {embed}: wall {wall:.2f} s; pressure "
      f"{st['ms_pressure'] / st['n_pressure_launches']:.1f} ms/launch (mean CG iterations {st['mean_n_cg']:.0f}), saturation "
      f"{st['ms_saturation'] / st['n_saturation_launches']:.1f} ms/launch (mean Nts {st['mean_nts']:.0f}); status ok: {not status.any()}")
