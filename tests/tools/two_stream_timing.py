"""Config 2 (1000 members, 128 x 128, fp64) as ONE plan on one stream against TWO plans of 500 members on two streams (two contexts):
members are independent, the kernels of the two halves interleave on the chip (the partial last round of one kernel -- 1000 members
are 3.9 rounds of 256 CUs -- is filled by the other half).   python tests/tools/two_stream_timing.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N, nT = 1000, bench.NTIME
model = bench.build_model(64, device=0)
perms = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
for parts in (1, 2, 4):
    ctxs = [_lib.Context.get(0)] + [_lib.Context(0) for _ in range(parts - 1)]
    bounds = np.linspace(0, N, parts + 1).astype(int)
    plans = []
    for c, lo, hi in zip(ctxs, bounds[:-1], bounds[1:]):
        p = ForwardPlan(model, hi - lo, bench.DT, nT, keep_history=True, ctx=c)
        p.set_inputs(perms[lo:hi], None, transformed=False)
        plans.append(p)
    for p in plans:
        p.run(0, nT)
    for p in plans:
        p.sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        # interleave the launches of the parts step by step so that neither stream runs far ahead of the other
        for k in range(nT):
            for p in plans:
                p.run(k, 1)
    for p in plans:
        p.sync()
    wall = (time.perf_counter() - t0) / reps
    outs = [p.outputs(want_wsats=False)[1] for p in plans]
    print(f"{parts} stream(s): {wall * 1e3:8.1f} ms per pass = {N * nT / wall:8.0f} ensemble-steps/s; prods checksum {float(np.concatenate(outs).sum()):.12f}", flush=True)
    for p in plans:
        p.close()
