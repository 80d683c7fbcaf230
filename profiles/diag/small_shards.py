"""Small member shards at 128 x 128 on one GPU (what one rank of an 8-GPU strong-scaled config 2 would hold): wall time of a 40-step forward
pass for N = 125 / 250 / 500 / 1000 members, fp64 and dtype = 32, as the library runs them (forward.default_blocks), and the per-kernel device
times of a one-block pass.     python profiles/diag/small_shards.py [dtype=64,32] [N=125,250,500,1000]
Implied strong-scaling factor for N_e = 1000 over G GPUs = rate(1000 / G members) * G / rate(1000 members): nothing here is a multi-GPU measurement."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from historymatching_amd.forward import BlockedForwardPlan, ForwardPlan, default_blocks  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

dts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,32").split(",")]
Ns = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "125,250,500,1000").split(",")]
perms_all = gaussian_fields_kron(128, 128, 2, 1, max(Ns), r=0.8, seed=1)
for dt_ in dts:
    model = bench.build_model(dt_, device=0)
    rates = {}
    for N in Ns:
        perms = perms_all[:N]
        nb = default_blocks(model, N)
        plan = BlockedForwardPlan(model, N, bench.DT, bench.NTIME, keep_history=False, blocks=nb)
        best = 1e9
        for rep in range(3):
            plan.set_inputs(perms, None, transformed=False)
            plan.sync()
            t0 = time.perf_counter()
            plan.run(0, bench.NTIME)
            plan.sync()
            best = min(best, time.perf_counter() - t0)
        plan.close()
        one = ForwardPlan(model, N, bench.DT, bench.NTIME, keep_history=False, device=0)
        for rep in range(2):
            one.set_inputs(perms, None, transformed=False)
            one.run(0, bench.NTIME)
            st = one.sync()
        one.close()
        rates[N] = N * bench.NTIME / best
        print(f"dtype {dt_}  N = {N:5d}  blocks {nb}: {1e3 * best:7.1f} ms per pass = {rates[N] / 1e3:6.2f} k ensemble-steps/s"
              f" ({rates[N] / N:6.1f} per member);  one block: pressure {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms, sweep "
              f"{st['ms_saturation'] / st['n_saturation_launches']:.3f} ms per launch; team_retries {st.get('team_retries', 0)}", flush=True)
    if 1000 in rates:
        for N in Ns:
            if 1000 % N == 0 and N != 1000:
                G = 1000 // N
                print(f"dtype {dt_}  implied strong scaling of N_e = 1000 over {G} GPUs (no communication in the forward model): "
                      f"{rates[N] * G / rates[1000]:.2f}x of the ideal {G}x  (throughput of a {N}-member shard: {rates[N] / rates[1000]:.2f} of the 1000-member one's)", flush=True)
