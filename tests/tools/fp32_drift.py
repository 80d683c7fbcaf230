"""fp32 forward mode against the fp64 forward mode over a whole run, same inputs, step by step.

    python tests/tools/fp32_drift.py <grid> <members> [steps=40] [sat_variant32=0] [seed=1]

Two plans without history (dtype 64 and dtype 32) advance side by side (tests/helpers.py:fp32_vs_fp64_drift); after every time step:
max, 99.9-percentile and mean |S32 - S64| over all members and cells, the largest producer-series difference so far, the water-in-place
difference (fraction of the pore volume; max / min over members) and whether the two modes took the same sub-step counts.  The last
line is the whole-run maximum (the bound DESIGN.md section 2 quotes; tests/test_configs_gpu.py asserts it)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.helpers import fp32_vs_fp64_drift  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sv32 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
print(f"# fp32 forward mode vs fp64 forward mode: {grid} x {grid}, {N} members (seed {seed}), sat_variant(dtype 32) = {sv32}")
print("# step  max|dS|   p99.9|dS|  mean|dS|   max|dprod|  water in place, fp64 - fp32 (max / min over members, fraction of pore volume)  same Nts  s")
t0 = time.time()


def report(r):
    print(f"{r['step']:5d}  {r['max']:.3e}  {r['p999']:.3e}  {r['mean']:.3e}  {r['prod']:.3e}   {r['wip_max']:+.3e} / {r['wip_min']:+.3e}   {r['same_nts']}  smax-1 {r['s_max'] - 1:+.2e}  "
          f"{time.time() - t0:.0f}{'' if not r['status'] else '   STATUS ' + str(r['status'])}", flush=True)


rows = fp32_vs_fp64_drift(grid, N, steps, seed=seed, sat_variant32=sv32, report=report)
print(f"# whole run: max|S32 - S64| = {max(r['max'] for r in rows):.3e}, p99.9 = {max(r['p999'] for r in rows):.3e}, max producer difference = "
      f"{max(r['prod'] for r in rows):.3e}, max |water-in-place difference| = {max(max(abs(r['wip_max']), abs(r['wip_min'])) for r in rows):.3e}, "
      f"max S32 - 1 = {max(r['s_max'] for r in rows) - 1:+.3e}")
