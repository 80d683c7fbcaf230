"""Timing of the large-grid path (CG pressure + generic saturation kernels) on shards of BASELINE configs 4 and 5.
   python tests/tools/large_grid_timing.py [n members nTime [pressure_variant [dtype [saturation_variant [embed]]]]]   (saturation_variant 5: tile teams instead of slabs at 256 wide)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import perms, wells_4corners  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402

cases = [(256, 64, 2), (512, 16, 1)] if len(sys.argv) < 4 else [tuple(int(a) for a in sys.argv[1:4])]
pressure_variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # 9: Jacobi-CG instead of the two-level preconditioner
dtype = int(sys.argv[5]) if len(sys.argv) > 5 else 64
sat_variant = int(sys.argv[6]) if len(sys.argv) > 6 else 0
embed = int(sys.argv[7]) if len(sys.argv) > 7 else 1           # 0: the generic kernels on the grid as given (no embedding in the next square)
for n, N, nTime in cases:
    gm = wells_4corners(ResSim(n, n, 2, 1, dtype=dtype))
    plan = ForwardPlan(gm, N, 0.025, nTime + 1, keep_history=False, device=0)
    plan.set_variant(pressure_variant, sat_variant)
    plan.set_debug("embed", embed)
    plan.set_inputs(perms(n, n, N, seed=3), None, transformed=False)
    plan.run(0, 1)  # first step: the lazily allocated solver buffers
    plan.sync()
    t0 = time.perf_counter()
    plan.run(1, nTime)
    st = plan.sync()
    wall = time.perf_counter() - t0
    _, _, status = plan.outputs(want_wsats=False)
    print(f"{n}x{n}, {N} members, {nTime} steps after one warm-up step, dtype {dtype}: wall {wall:.2f} s; pressure {st['ms_pressure'] / st['n_pressure_launches']:.1f} ms/launch "
          f"(mean CG iterations {st['mean_n_cg']:.0f}), saturation {st['ms_saturation'] / st['n_saturation_launches']:.1f} ms/launch "
          f"(mean Nts {st['mean_nts']:.0f}); status ok: {not status.any()}")
    plan.close()
