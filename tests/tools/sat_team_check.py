"""Multi-tile saturation sweep (sat128t.hip: teams of workgroups) against the tiled single-workgroup kernel (sat_variant 3):
bit-identical saturations / producer series / sub-step counts, and launch averages.
   python tests/tools/sat_team_check.py [nx ny members nTime [dtype]]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import perms, wells_4corners  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

cases = [(256, 256, 5, 2), (256, 128, 3, 2), (128, 256, 3, 2), (512, 512, 2, 1), (256, 256, 70, 1), (256, 256, 64, 2), (512, 512, 16, 1)]
if len(sys.argv) >= 5:
    cases = [tuple(int(a) for a in sys.argv[1:5])]
dtype = int(sys.argv[5]) if len(sys.argv) > 5 else 64
for nx, ny, N, nTime in cases:
    gm = wells_4corners(ResSim(nx, ny, 2, 1, dtype=dtype))
    x = perms(nx, ny, N, seed=3)
    out = {}
    for v in (0, 3):
        plan = ForwardPlan(gm, N, 0.025, nTime, keep_history=True, device=0)
        plan.set_variant(0, v)
        plan.set_inputs(x, None, transformed=False)
        plan.run()
        st = plan.sync()
        w, p, status = plan.outputs()
        out[v] = (w, p, status, st["ms_saturation"] / st["n_saturation_launches"], st["mean_nts"])
        plan.close()
    same = np.array_equal(out[0][0], out[3][0]) and np.array_equal(out[0][1], out[3][1])
    print(f"{nx}x{ny}, {N} members, {nTime} steps: teams {out[0][3]:.1f} ms/launch, tiled {out[3][3]:.1f} ms/launch; mean Nts {out[0][4]:.0f}/{out[3][4]:.0f}; "
          f"status {out[0][2].max()}/{out[3][2].max()}; bit-identical: {same}", flush=True)
    if not same:
        d = np.abs(out[0][0] - out[3][0])
        print("   max |diff|", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
