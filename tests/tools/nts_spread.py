"""Spread of the CFL sub-step count over the members of an ensemble (it decides how much a round of workgroup teams waits for its
slowest member).  python tests/tools/nts_spread.py [n members steps]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import perms, wells_4corners  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

n, N, steps = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 128, 3)))
gm = wells_4corners(ResSim(n, n, 2, 1))
plan = ForwardPlan(gm, N, 0.025, steps, keep_history=False, device=0)
plan.set_inputs(perms(n, n, N, seed=3), None, transformed=False)
plan.run()
plan.sync()
nts = plan.get_field("nts").reshape(N, steps)
for k in range(steps):
    v = nts[:, k]
    rounds = [v[i:i + 64] for i in range(0, N, 64)]
    waste = 1 - v.sum() / sum(r.max() * len(r) for r in rounds)
    print(f"{n}x{n} step {k}: Nts min {v.min()} mean {v.mean():.0f} max {v.max()}; rounds of 64 wait for their slowest member: {100 * waste:.1f} % idle")
plan.close()
