"""The larger grids as member blocks on streams of their own (forward.BlockedForwardPlan with explicit bounds): config 4's shard (512 members, 256 x 256,
fp64) and config 5's (125 members, 512 x 512, dtype = 32), wall time of a 10-step run after a warm-up run, sweeps' time-out retries counted.
     python profiles/diag/blocks_time_large.py [256|512] [dtype]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import perms, wells_4corners  # noqa: E402
from historymatching_amd.forward import BlockedForwardPlan  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, dtype = (512, 64) if n == 256 else (125, 32)
if len(sys.argv) > 2:
    dtype = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gm = wells_4corners(ResSim(n, n, 2, 1, dtype=dtype))
x = perms(n, n, N, seed=3)
ref = None
for nb in ((1, 2, 3) if steps <= 10 else (1, 2)):
    bounds = np.linspace(0, N, nb + 1).astype(int)
    plan = BlockedForwardPlan(gm, N, 0.025, steps, keep_history=False, bounds=bounds)
    best = 1e9
    for rep in range(2):
        plan.set_inputs(x, None, transformed=False)
        plan.sync()
        t0 = time.perf_counter()
        plan.run(0, steps)
        st = plan.sync()
        best = min(best, time.perf_counter() - t0)
    S, p, status = plan.outputs()
    if ref is None:
        ref = (S, p)
    print(f"{n} x {n}, dtype {dtype}, {N} members, {nb} block(s): {1e3 * best / steps:8.2f} ms per time step; status ok {not status.any()}; team retries {st['team_retries']}, slab redos {st['slab_redos']}, "
          f"handed to the CG {st['nd_fallbacks']}; identical to one block: {np.array_equal(S, ref[0]) and np.array_equal(p, ref[1])}", flush=True)
    plan.close()
