"""How much of the grid is still exactly dry (S == 0) per time step, at the granularity of the nested dissection's level-8 subtrees
(8 x 8 cells plus a ring of one cell): the fraction of subtrees whose pressure fronts see the same coefficients as the step before.
    python tests/tools/dry_fraction.py [members=8]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, nT = 128, 40
_, gm = make_models(n, n)
plan = ForwardPlan(gm, M, 0.025, nT, keep_history=True, device=0)
plan.set_inputs(perms(n, n, M, seed=1), None, transformed=False)
plan.run()
w, _, status = plan.outputs()
plan.close()
w = w.reshape(M, nT + 1, n, n)
fr8, fr32 = [], []
for k in range(nT):
    wet = np.pad(w[:, k] != 0.0, ((0, 0), (1, 1), (1, 1)))
    # a block is clean at step k (k >= 1) if block + ring is dry at steps k and k - 1 (S is checked at k: dry now implies dry before)
    def clean(b):
        nb = n // b
        c = np.ones((M, nb, nb), bool)
        for i in range(nb):
            for j in range(nb):
                c[:, i, j] = ~wet[:, i * b:i * b + b + 2, j * b:j * b + b + 2].any(axis=(1, 2))
        return c.mean()
    fr8.append(clean(8))
    fr32.append(clean(32))
print("time step:            " + " ".join(f"{k:4d}" for k in range(0, nT, 3)))
print("clean 8 x 8 blocks:   " + " ".join(f"{fr8[k]:4.2f}" for k in range(0, nT, 3)))
print("clean 32 x 32 blocks: " + " ".join(f"{fr32[k]:4.2f}" for k in range(0, nT, 3)))
print(f"mean over the run: 8 x 8 {np.mean(fr8[1:]):.3f}, 32 x 32 {np.mean(fr32[1:]):.3f}; status ok {not status.any()}")
