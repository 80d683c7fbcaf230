"""fp32 forward mode against the fp64 forward mode over a whole run, same inputs, step by step.

    python tests/tools/fp32_drift.py <grid> <members> [steps=40] [sat_variant32=0] [seed=1]

Two plans without history (dtype 64 and dtype 32) advance side by side; after every time step the saturations are read back and
compared: max and 99.9-percentile |S32 - S64| over all members and cells, the largest producer-series difference, the
water-in-place deficit of both modes (injected volume minus water in place minus water produced is not tracked here: the deficit is
measured against the fp64 plan's water in place, per member, as a fraction of the pore volume), and whether the two modes took the
same sub-step counts.  One line per time step; the last line is the whole-run maximum (the bound DESIGN.md section 2 quotes)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from tests.helpers import make_models, perms  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sv32 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
DT, NTIME = 0.025, 40

_, g64 = make_models(grid, grid, dtype=64)
_, g32 = make_models(grid, grid, dtype=32)
x = perms(grid, grid, N, seed=seed)
p64 = ForwardPlan(g64, N, DT, NTIME, keep_history=False)
p32 = ForwardPlan(g32, N, DT, NTIME, keep_history=False)
p32.set_variant(0, sv32)
p64.set_inputs(x, None, transformed=False)
p32.set_inputs(x, None, transformed=False)
print(f"# fp32 forward mode vs fp64 forward mode: {grid} x {grid}, {N} members (seed {seed}), sat_variant(dtype 32) = {sv32}")
print("# step  max|dS|   p99.9|dS|  mean|dS|   max|dprod|  water-in-place deficit (max over members, fraction of pore volume)  same Nts  s")
worst = dict(S=0.0, p999=0.0, prod=0.0, wip=0.0)
t0 = time.time()
for k in range(steps):
    p64.run(k, 1)
    p32.run(k, 1)
    S64 = p64.get_field("S").reshape(N, -1)
    S32 = p32.get_field("S").reshape(N, -1).astype(np.float64)
    d = np.abs(S32 - S64)
    wip = (S64.mean(axis=1) - S32.mean(axis=1))  # mean saturation = water in place / pore volume
    _, pr64, st64 = p64.outputs(want_wsats=False)
    _, pr32, st32 = p32.outputs(want_wsats=False)
    dp = np.abs(pr32[:, k].astype(np.float64) - pr64[:, k]).max()
    nts_same = bool(np.array_equal(p64.get_field("nts")[:, k], p32.get_field("nts")[:, k]))
    # the 99.9 percentile over a sample (np.partition on the whole array at 512 x 512 x 125 is slow but fine)
    flat = d.reshape(-1)
    kth = int(0.999 * (flat.size - 1))
    p999 = float(np.partition(flat, kth)[kth])
    worst["S"] = max(worst["S"], float(d.max()))
    worst["p999"] = max(worst["p999"], p999)
    worst["prod"] = max(worst["prod"], float(dp))
    worst["wip"] = max(worst["wip"], float(np.abs(wip).max()))
    print(f"{k + 1:5d}  {d.max():.3e}  {p999:.3e}  {d.mean():.3e}  {dp:.3e}   {wip.max():+.3e} / {wip.min():+.3e}   {nts_same}  {time.time() - t0:.0f}"
          f"{'' if not (st64.any() or st32.any()) else '   STATUS ' + str(int(st64.max())) + '/' + str(int(st32.max()))}", flush=True)
print(f"# whole run: max|S32 - S64| = {worst['S']:.3e}, p99.9 = {worst['p999']:.3e}, max producer difference = {worst['prod']:.3e}, "
      f"max water-in-place deficit = {worst['wip']:.3e}")
p64.close()
p32.close()
