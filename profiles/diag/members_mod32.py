"""Is a one-block 128 x 128 forward pass sensitive to the member count modulo 32?  Device time of the pressure step and of the sweep per launch and
per member, and the wall time of a 40-step pass, for a list of member counts.     python profiles/diag/members_mod32.py [N,N,...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "320,328,334,352,360,992,1000,1024").split(",")]
perms_all = gaussian_fields_kron(128, 128, 2, 1, max(Ns), r=0.8, seed=1)
model = bench.build_model(64, device=0)
for N in Ns:
    plan = ForwardPlan(model, N, bench.DT, bench.NTIME, keep_history=False, device=0)
    best, bp, bs = 1e9, 1e9, 1e9
    for rep in range(3):
        plan.set_inputs(perms_all[:N], None, transformed=False)
        plan.sync()
        t0 = time.perf_counter()
        plan.run(0, bench.NTIME)
        st = plan.sync()
        best = min(best, time.perf_counter() - t0)
        bp = min(bp, st["ms_pressure"] / st["n_pressure_launches"])
        bs = min(bs, st["ms_saturation"] / st["n_saturation_launches"])
    plan.close()
    print(f"N = {N:5d} (mod 32 = {N % 32:2d}): pass {1e3 * best:7.1f} ms = {1e6 * best / N / bench.NTIME:6.3f} us per member-step; pressure {bp:6.3f} ms = "
          f"{1e3 * bp / N:6.3f} us/member; sweep {bs:6.3f} ms = {1e3 * bs / N:6.3f} us/member", flush=True)
