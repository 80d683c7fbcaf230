"""PCIe-inclusive cost of the drop-in call at BASELINE config 2: forward_model(perms) -> [wsats (N, 41, Nxy), prods] through
hm_forward_batched (host buffers in, full saturation history out) against the device time of the same run."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from historymatching_amd.forward import make_forward_model  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
model = bench.build_model(64, device=0)
x = gaussian_fields_kron(128, 128, 2, 1, N, r=0.8, seed=1)
for hist in (True, False):
    fm = make_forward_model(model, bench.DT, bench.NTIME, return_history=hist)
    t0 = time.perf_counter()
    fm(x)  # first call: builds the device plan (22 GB of buffers)
    first = time.perf_counter() - t0
    wall = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        w, p = fm(x)
        wall = min(wall, time.perf_counter() - t0)
    st = model.last_stats
    print(f"return_history={hist}: first call {first:.2f} s, best-of-3 call wall {wall:.2f} s, device {st['ms_total'] / 1e3:.2f} s, output {w.nbytes / 1e9:.2f} GB -> "
          f"{N * bench.NTIME / wall:.0f} ensemble-steps/s PCIe-inclusive")
