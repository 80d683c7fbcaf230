"""Config 2's forward pass as B member blocks on B streams of one GPU (bench.py's `two_streams` leg is B = 2):
   python tests/tools/multi_stream_timing.py [B ...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = 1000
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    bounds = np.linspace(0, N, B + 1).astype(int)
    ctxs = [_lib.Context.get(0)] + [_lib.Context(0) for _ in range(B - 1)]
    plans = []
    for c, lo, hi in zip(ctxs, bounds[:-1], bounds[1:]):
        p = ForwardPlan(model, hi - lo, bench.DT, bench.NTIME, keep_history=True, ctx=c)
        p.set_inputs(x[lo:hi], None, transformed=False)
        plans.append(p)

    def run(reps):
        for _ in range(reps):
            for k in range(bench.NTIME):
                for p in plans:
                    p.run(k, 1)
        for p in plans:
            p.sync()

    run(1)
    t0 = time.perf_counter()
    run(2)
    wall = (time.perf_counter() - t0) / 2
    print(f"{B} block(s) of {N // B} members on {B} stream(s): {1e3 * wall:.1f} ms per pass -> {N * bench.NTIME / wall:.0f} ensemble-steps/s", flush=True)
    for p in plans:
        p.close()
