"""The CPU baselines SURVEY.md 8d asks for beside the headline one (bench.py times config 2's forward model):
  * BASELINE config 1 (N_e=100, 20x20, nTime=40): oracle forward model with 1 process and with one process per core,
    against the drop-in forward_model on the GPU (host arrays in and out);
  * ensemble-smoother update at config 3's shape (N=1000, M=16384, n_obs=160): oracle ens_update0 with all BLAS threads
    against the GPU plan (device time and host-array wall time, fp64 and fp32);
  * localised update: oracle ens_update0_loc with 1 BLAS thread on a sample of state elements (scaled to M), against the GPU.
Run on the GPU box: python tests/tools/cpu_baselines_extra.py   (each section runs in a process of its own: a forked pool and a
many-thread BLAS do not mix)"""
import os
import sys
import time
from pathlib import Path

import numpy as np
from threadpoolctl import threadpool_limits

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import oracle_sim_and_noise, wells_4corners  # noqa: E402
from historymatching_amd.forward import make_forward_model  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402
from historymatching_amd.localization import bump, pairwise_distances  # noqa: E402
from historymatching_amd.ressim import ResSim  # noqa: E402
from historymatching_amd.update import UpdatePlan  # noqa: E402
from oracle import es as oes  # noqa: E402
from oracle.ressim import ResSim as OResSim, default_wells, forward_model as oracle_forward, make_pool  # noqa: E402

ncore = os.cpu_count() or 1
nblas = min(ncore, 64)
DT, NT = 0.025, 40
if len(sys.argv) < 2:
    import subprocess

    for sec in ("c1", "upd", "loc"):
        subprocess.run([sys.executable, __file__, sec], check=False)
    sys.exit(0)
section = sys.argv[1]

def config1():
    # ---- config 1
    N, n = 100, 20
    x = gaussian_fields_kron(n, n, 2, 1, N, r=0.8, seed=1)
    om = default_wells(OResSim(n, n, 2, 1))
    with make_pool(min(ncore, N)) as pool:   # the process pool first: fork before this process touches the GPU
        oracle_forward(om, x[: min(ncore, N)], None, DT, 1, pool=pool)
        t0 = time.perf_counter(); oracle_forward(om, x, None, DT, NT, pool=pool); cpuN = time.perf_counter() - t0
    t0 = time.perf_counter(); ow, op = oracle_forward(om, x, None, DT, NT, pool=None); cpu1 = time.perf_counter() - t0
    gm = wells_4corners(ResSim(n, n, 2, 1))
    fm = make_forward_model(gm, DT, NT)
    fm(x)
    t0 = time.perf_counter(); w, p = fm(x); gpu = time.perf_counter() - t0
    print(f"config 1 (N_e={N}, {n}x{n}, nTime={NT}): GPU drop-in {gpu * 1e3:.1f} ms = {N * NT / gpu:.0f} ensemble-steps/s; oracle 1 process {cpu1:.2f} s = "
          f"{N * NT / cpu1:.0f}/s; oracle {min(ncore, N)} processes {cpuN:.2f} s = {N * NT / cpuN:.0f}/s", flush=True)
    err = np.abs(w - ow).reshape(N, -1).max(1)
    worst = int(err.argmax())
    _, noise = oracle_sim_and_noise(om, x[worst], DT, NT)
    print(f"   max |GPU - oracle| over members: median {np.median(err):.1e}, worst member {worst}: {err[worst]:.1e} (the oracle's own solver noise for that member: {noise:.1e})", flush=True)



N, M, n_obs = 1000, 128 * 128, 160
rng = np.random.RandomState(0)
E, Y, obs = rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs)
pert, decorr = 0.1 * rng.randn(N, n_obs), 3.0 * np.eye(n_obs)


def update():
    with threadpool_limits(limits=nblas):
        oes.ens_update0(E[:, :256], Y, obs, pert, decorr)
        t0 = time.perf_counter(); ref = oes.ens_update0(E, Y, obs, pert, decorr); cpu_upd = time.perf_counter() - t0
    for dtype in (64, 32):
        plan = UpdatePlan(N, N, M, n_obs, dtype=dtype)
        t0 = time.perf_counter()
        plan.set_inputs(E, Y, obs, pert, decorr)
        st = plan.run_local()
        out = plan.output()
        wall = time.perf_counter() - t0
        ms = min(plan.run_local()["ms_update"] for _ in range(5))
        err = np.abs(out - ref).max() / np.abs(ref - E).max()
        print(f"ens_update0 N={N} M={M} n_obs={n_obs}: oracle ({nblas} BLAS threads, literal order) {cpu_upd:.2f} s; GPU fp{dtype}: device {ms:.3f} ms, host arrays in/out "
              f"{wall * 1e3:.0f} ms; max err / max |increment| = {err:.1e}", flush=True)
        plan.close()



def localised():
    # ---- localised update
    model = ResSim(128, 128, 2, 1)
    near01 = np.array([0.12, 0.87])
    model.prd_xy = [[a, b] for b in model.Ly * near01 for a in model.Lx * near01]
    xy_obs = np.tile(model.ind2xy(model.xy2ind(*model.prd_xy.T)), NT)
    xy_prm = model.ind2xy(np.arange(model.Nxy))
    taper = bump(pairwise_distances(xy_prm.T, xy_obs.T) / 1.2)
    sample = np.arange(0, M, M // 128)[:128]
    with threadpool_limits(limits=1):
        t0 = time.perf_counter(); refl = oes.ens_update0_loc(E[:, sample], Y, obs, pert, decorr, taper[sample]); cpu_loc = time.perf_counter() - t0
    for dtype in (64, 32):
        plan = UpdatePlan(N, N, M, n_obs, dtype=dtype, localized=True)
        plan.set_inputs(E, Y, obs, pert, decorr, taper=taper)
        plan.run_local()
        ms = min(plan.run_local()["ms_update"] for _ in range(3))
        out = plan.output()
        err = np.abs(out[:, sample] - refl).max() / np.abs(refl - E[:, sample]).max()
        print(f"ens_update0_loc: oracle (1 BLAS thread) {cpu_loc / len(sample) * 1e3:.1f} ms per state element -> {cpu_loc / len(sample) * M:.0f} s for M={M} on one core, "
              f"{cpu_loc / len(sample) * M / ncore:.1f} s if spread over {ncore} cores; GPU fp{dtype}: {ms:.2f} ms; max err / max |increment| on the sample = {err:.1e}", flush=True)
        plan.close()


{'c1': config1, 'upd': update, 'loc': localised}[section]()
