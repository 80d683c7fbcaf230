#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ensemble forward model.

Metric (BASELINE.json): ensemble-steps/sec = N_e * nTime / wall(forward_model), workload = config[1]:
N_e=1000 members, 128x128 grid, forward model only (TPFA pressure solve + explicit upwind saturation sweep per
member per step), fp64, nTime=40, wells/fluid/dt of notebooks/HistoryMatch.py:97,177-190,219-221.

A "step" (--steps) is ONE pass of the hot path over the whole batch: all N_e members advanced nTime=40 time
steps, inputs (permeability, initial saturation) already resident in HBM.  Multi-GPU (--gpus N, launched by
torch.distributed.run, which only provides RANK / LOCAL_RANK / WORLD_SIZE -- no PyTorch is imported here): members are
independent, so every rank runs its own N_e members with no data-path collective ("weak" scaling); value = total
member-steps of all ranks / max-over-ranks time, bracketed by a barrier + device synchronisation on both sides.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library on the launch
stream) and `cpu_baseline` (the NumPy/SciPy oracle on the host cores, bounded sample, rank 0 at N=1 only).
Beside `value` (outside its timed region): `es_update` (the analysis step at config 3's shape; 4 ES-MDA passes of config 3;
at N > 1 the analysis step row-sharded over the ranks with RCCL), `config4` (BASELINE config 4: N_e = 4096 at
256 x 256 split over the ranks -- strong scaling -- one ES-MDA pass = forward model + analysis step over RCCL) and, from 4 ranks
up or with --config5, `config5` (N_e = 1000 at 512 x 512, localised analysis with column-sharded solves).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NX = NY = 128
N_E = 1000
NTIME = 40
DT = 0.025
# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, fp64 VALU = 16 lanes / cycle / SIMD, 2.4 GHz; fp64 vector peak = fp64 matrix peak
HBM_PEAK_GBS = 8000.0
CLOCK_HZ = 2.4e9
N_CU = 256
ND_FACTOR_DOUBLES, ND_ARENA_DOUBLES = 629312, 612620  # per member: hm_debug_nd_tables info[2], info[3] at 128 x 128
DP_LANE_RATE = N_CU * 4 * 16 * CLOCK_HZ          # double-precision lane-instructions / s (39.3e12)
FP64_PEAK_TFLOPS = 2 * DP_LANE_RATE / 1e12        # 78.6: one FMA per lane-slot
FP32_MATRIX_PEAK_TFLOPS = 157.3


def build_model(dtype=64, device=None, n=NX):
    from historymatching_amd.ressim import ResSim

    m = ResSim(n, n, 2, 1, dtype=dtype, device=device)
    near01 = np.array([0.12, 0.87])  # HistoryMatch.py:177-190
    m.prd_xy = [[x, y] for y in m.Ly * near01 for x in m.Lx * near01]
    m.inj_xy = [[m.Lx / 2, m.Ly / 2]]
    m.inj_rates = [[1]]
    m.prd_rates = np.ones((4, 1)) / 4
    return m


def cpu_baseline(members, steps, nproc):
    """Oracle (oracle/ressim.py, NumPy + SciPy spsolve) on a bounded sample of the same workload, parallelised the
    way the reference does it (utils.py:201-224: one process per core, BLAS pinned to 1 thread).  The pool is
    created and warmed before the timed region (fork/import cost is not simulator time)."""
    from historymatching_amd.geostat import gaussian_fields_kron
    from oracle.ressim import ResSim, default_wells, forward_model, make_pool

    om = default_wells(ResSim(NX, NY, 2, 1))
    x = gaussian_fields_kron(NX, NY, 2, 1, members, r=0.8, seed=12345)
    with make_pool(nproc) as pool:
        forward_model(om, x, None, DT, 1, pool=pool)  # warm-up: every worker runs one step
        t0 = time.perf_counter()
        forward_model(om, x, None, DT, steps, pool=pool)
        wall = time.perf_counter() - t0
    return members * steps / wall, wall


def config1_reference_run(device, nproc):
    """BASELINE config 1 -- "Reference HistoryMatch.py CPU run: N_e=100, default 2D grid, 1 ES-MDA iteration" -- both ways: the oracle
    on the host cores parallelised like the reference (utils.py:201-224: one process per core) and the same forward run + one analysis
    step on the GPU.  Default grid and wells of HistoryMatch.py:97, 177-190 (20 x 20 cells on 2 x 1), nTime = 40."""
    import scipy.linalg as sla

    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.geostat import gaussian_fields_kron
    from historymatching_amd.update import UpdatePlan
    from oracle import es as oes
    from oracle.ressim import ResSim, default_wells, forward_model, make_pool

    n, N = 20, 100
    x = gaussian_fields_kron(n, n, 2, 1, N, r=0.8, seed=77)
    R12 = reference_obs_error(160)
    rng = np.random.RandomState(5)
    om = default_wells(ResSim(n, n, 2, 1))
    procs = max(1, min(nproc, N))
    with make_pool(procs) as pool:
        forward_model(om, x[:procs], None, DT, 1, pool=pool)  # warm-up
        t0 = time.perf_counter()
        _, prods_cpu = forward_model(om, x, None, DT, NTIME, pool=pool)
        cpu_fwd = time.perf_counter() - t0
    obs = np.clip(prods_cpu[0].reshape(-1) + R12 @ rng.randn(160), 0, 1)
    perturbs, decorr = rng.randn(N, 160) @ R12.T, sla.inv(R12.T)
    t0 = time.perf_counter()
    post_cpu = oes.ens_update0(x, prods_cpu.reshape(N, -1), obs, perturbs, decorr)
    cpu_upd = time.perf_counter() - t0
    model = build_model(64, device=device, n=n)
    plan = ForwardPlan(model, N, DT, NTIME, keep_history=False, device=device)
    upd = UpdatePlan(N, N, n * n, 160, dtype=64, device=device)
    for _ in range(2):  # second pass timed
        plan.set_inputs(x, None, transformed=False)
        t0 = time.perf_counter()
        plan.run()
        st = plan.sync()
        gpu_fwd = time.perf_counter() - t0
    _, prods_gpu, status = plan.outputs(want_wsats=False)
    upd.set_inputs(x, prods_gpu.reshape(N, -1), obs, perturbs, decorr)
    upd.run_local()
    t0 = time.perf_counter()
    su = upd.run_local()
    gpu_upd = time.perf_counter() - t0
    post_gpu = upd.output()
    plan.close()
    upd.close()
    return {"workload": "N_e=100, 20x20 (HistoryMatch.py:97), nTime=40, one forward run + one ens_update0 (fp64)",
            "cpu": {"kind": "port", "cores": procs, "forward_s": cpu_fwd, "update_s": cpu_upd, "ensemble_steps_per_s": N * NTIME / cpu_fwd},
            "gpu": {"forward_s": gpu_fwd, "forward_device_ms": st["ms_total"], "update_s": gpu_upd, "update_device_ms": su["ms_update"],
                    "ensemble_steps_per_s": N * NTIME / gpu_fwd, "status_ok": bool(not status.any())},
            "max_abs_diff_producer_series": float(np.abs(prods_gpu - prods_cpu).max()),
            "max_abs_diff_posterior": float(np.abs(post_gpu - post_cpu).max())}


def nd_mfma_count():
    """v_mfma_f64_16x16x4 instructions per member and time step of the nested-dissection factorisation (press_nd.hip), from the
    symbolic tables: a front with st pivot tiles and bt boundary tiles runs, per pivot tile p with kreg 4-row groups, kreg matrix
    instructions for every tile it forms or updates: W^T tiles (T - p - 1), later panel tiles, and the bt (bt + 1) / 2 trailing tiles."""
    import ctypes as C

    from historymatching_amd import _lib

    lib = _lib.load()
    info = (C.c_longlong * 64)()
    _lib.check(lib.hm_debug_nd_tables(NX, NY, info, None, None, None, None), "hm_debug_nd_tables")
    fronts = np.zeros((int(info[0]), int(info[19])), dtype=np.int32)
    _lib.check(lib.hm_debug_nd_tables(NX, NY, info, fronts.ctypes.data_as(C.POINTER(C.c_int)), None, None, None), "hm_debug_nd_tables")
    total = 0
    for F in fronts:
        if int(F[0]) == 10:  # the leaves are eliminated one per lane by a banded L D L^T on the vector pipe (k_nd_leaf)
            continue
        b, st, bt, kreg_last = int(F[2]), int(F[3]), int(F[4]), int(F[10])
        T = st + bt
        ntrail = bt * (bt + 1) // 2 if b > 0 else 0
        for p in range(st):
            kreg = kreg_last if p == st - 1 else 4
            later_panel = sum(T - q for q in range(p + 1, st))
            total += kreg * ((T - p - 1) + later_panel + ntrail)
    return total


def reference_obs_error(n_obs):
    """R12 of the reference's observation-error model (HistoryMatch.py:243-259: 1e-2 * toeplitz(exp(-k/2), entries below
    1e-2 cut) per well, kron I_nPrd, lower Cholesky factor)."""
    from historymatching_amd.obs import obs_error_model

    return obs_error_model(n_obs // 4, 4)[1]


def es_update_timing(device):
    """BASELINE.json's second metric (ES-MDA update wall-time): device time of one analysis step (HIP events inside the library),
    inputs resident in HBM, fp32 state contractions on the matrix cores, every N x n_obs quantity in fp64, the reference's
    correlated observation error; flops by SURVEY.md 8d (min-flop order).  Headline = config 3's shape (N=1000 members, M=128*128
    state elements, n_obs=160); the same step at config 4's and config 5's ensemble/state sizes beside it (`by_shape`)."""
    import scipy.linalg as sla

    from historymatching_amd import _lib
    from historymatching_amd.update import UpdatePlan

    n_obs = 160
    R12 = reference_obs_error(n_obs)
    decorr = sla.inv(R12.T)
    rng = np.random.default_rng(0)
    shapes = [(N_E, NX * NY, "config 3: N_e=1000, 128x128"), (4096, 256 * 256, "config 4: N_e=4096, 256x256 (whole ensemble on one GPU)"),
              (1000, 512 * 512, "config 5: N_e=1000, 512x512 (global analysis)")]
    out, by_shape = None, []
    for N, M, label in shapes:
        plan = UpdatePlan(N, N, M, n_obs, dtype=32, device=device)
        plan.set_inputs(rng.standard_normal((N, M), dtype=np.float32), rng.random((N, n_obs)), rng.random(n_obs),
                        rng.standard_normal((N, n_obs)) @ R12.T, decorr)
        plan.run_local()  # warm-up (also forms R from decorr: done once per decorr, i.e. once per ES-MDA assimilation)
        ms = sorted(plan.run_local()["ms_update"] for _ in range(5))   # one step at a time, host synchronisation after each
        # Steps queued back to back, as ES-MDA chains them behind the forward model (no idle gap in front of a step's first kernel):
        # 0.1 s of untimed batches, then fifteen batches of ten, the median batch.  After an idle spell (the host-synchronised steps
        # above, the set-up of the inputs) the device needs 15-20 ms of work -- on some boxes of this pool several times that -- to
        # get back to the clocks it holds behind a forward pass: first batches 0.171 ms, settled 0.158 ms at config 3's shape
        # (profiles/diag/upd_ramp.py, upd_after_forward.py).
        reps, batches = 10, []
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.1:
            for _ in range(reps):
                _lib.check(plan.lib.hm_upd_run(plan.h), "hm_upd_run")
            plan.sync()
        for _ in range(int(os.environ.get("HM_BENCH_UPD_BATCHES", "15"))):
            for _ in range(reps):
                _lib.check(plan.lib.hm_upd_run(plan.h), "hm_upd_run")
            batches.append(plan.sync()["ms_update"] / reps)
        med = sorted(batches)[len(batches) // 2]
        plan.close()
        flops = 4.0 * N * n_obs * M
        entry = {"shape": label, "N": N, "M": M, "n_obs": n_obs, "wall_ms": med, "first_batch_ms": batches[0], "isolated_step_median_ms": ms[len(ms) // 2],
                 "tflops": flops / (med * 1e-3) / 1e12, "mfma_frac_of_fp32_peak": flops / (med * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS}
        by_shape.append(entry)
        if out is None:
            out = {"wall_ms": med, "isolated_step_median_ms": ms[len(ms) // 2], "best_ms": ms[0],
                   "timing": "device time (HIP events) per analysis step, 10 steps queued back to back; median of 15 such batches behind 0.1 s of untimed batches (the clocks of a busy device)",
                   "batches_ms": batches,
                   "config": f"N={N}, M={M}, n_obs={n_obs}, fp32 (config 3 shape), correlated R of HistoryMatch.py:243-259",
                   "flops_min_order": flops, "tflops": entry["tflops"], "mfma_peak_tflops": FP32_MATRIX_PEAK_TFLOPS,
                   "mfma_frac_of_fp32_peak": entry["mfma_frac_of_fp32_peak"]}
    out["by_shape"] = by_shape
    return out


def es_update_sharded_timing(device, comm, reps=5):
    """The analysis step row-sharded over the ranks (weak scaling like the forward metric: every rank holds
    N_e=1000 members of an N_e x world ensemble, M=128*128, n_obs=160, fp32 state): `hm_upd_run_comm` = three local phases with
    two RCCL all-reduces issued by the library on its own device buffers in between (column sums; X^T S and S^T S), all in stream
    order.  Wall time per update, barrier + device synchronisation on both sides, maximum over ranks.  Called by EVERY rank."""
    import scipy.linalg as sla

    from historymatching_amd.dist import sharded_update
    from historymatching_amd.update import UpdatePlan

    world = comm.world_size
    N, M, n_obs = N_E, NX * NY, 160
    rng = np.random.RandomState(100 + comm.rank)
    R12 = reference_obs_error(n_obs)
    plan = UpdatePlan(N * world, N, M, n_obs, dtype=32, device=device)
    plan.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), np.random.RandomState(7).rand(n_obs), rng.randn(N, n_obs) @ R12.T, sla.inv(R12.T))
    sharded_update(plan, comm, fetch=False)  # warm-up
    comm.barrier()
    t0 = time.perf_counter()
    dev_ms = comm_ms = 0.0
    for _ in range(reps):
        st = sharded_update(plan, comm, fetch=False)  # ends with a stream synchronisation
        dev_ms += st["ms_update"]
        comm_ms += st.get("ms_comm", 0.0)
    comm.barrier()
    wall = comm.all_reduce_max((time.perf_counter() - t0) / reps)
    nbytes = sum(plan.reduce_buffer(w)[1] * np.dtype(plan.reduce_buffer(w)[2]).itemsize for w in (0, 1, 2, 3))
    plan.close()
    return {"wall_ms": 1e3 * wall, "device_ms_phases": dev_ms / reps, "device_ms_collectives": comm_ms / reps, "n_ranks": world,
            "members_total": N * world, "members_per_rank": N, "M": M, "n_obs": n_obs, "allreduce_bytes_per_update": int(nbytes),
            "collective": ("RCCL all-reduce issued by the library in place on its device buffers (hm_upd_run_comm)" if comm.rccl is not None
                           else f"host channel (RCCL unavailable: {comm.rccl_error})"),
            "config": "row-sharded global analysis step, fp32 state contractions on the matrix cores, weak scaling"}


def es_mda_c3(device, perms, n_iter=4):
    """BASELINE.json config 3: N_e=1000, 128x128, 4 ES-MDA passes (forward model in fp32 mode + fp32 matrix-core analysis), the
    ensemble resident in HBM throughout (update.es_mda_device), observation error as the reference's (correlated in time).
    Observations = member 0's simulated production + noise."""
    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.update import es_mda_device

    model = build_model(32, device=device)  # config 3 is the fp32 configuration: fp32 saturation sweep (sat32s, compensated state), fp64 pressure
    n_obs = NTIME * 4
    rng = np.random.RandomState(4)
    R12 = reference_obs_error(n_obs)
    fwd = ForwardPlan(model, 1, DT, NTIME, keep_history=False, device=device)
    fwd.set_inputs(perms[:1], None, transformed=False)
    fwd.run()
    fwd.sync()
    truth_obs = fwd.outputs(want_wsats=False)[1].reshape(-1)
    fwd.close()
    obs = np.clip(truth_obs + R12 @ rng.randn(n_obs), 0, 1)
    st = {}
    t0 = time.perf_counter()
    post = es_mda_device(model, perms, obs, R12, DT, NTIME, n_iter=n_iter, rng=rng, dtype=32, device=device, stats=st)
    wall = time.perf_counter() - t0
    # the assimilation must have pulled the ensemble's simulated production towards the observations
    return {"iterations": n_iter, "wall_s": wall, "device_ms_forward": st["ms_forward"], "device_ms_update": st["ms_update"],
            "fallbacks": {k: int(st.get(k, 0)) for k in FALLBACK_KEYS},
            "ensemble_steps_per_s_incl_updates": len(perms) * NTIME * n_iter / wall,
            "posterior_finite": bool(np.isfinite(post).all()),
            "rms_change_of_log_perm": float(np.sqrt(np.mean((post - perms) ** 2))),
            "config": f"N_e={len(perms)}, {NX}x{NY}, {n_iter} ES-MDA passes, forward model dtype=32 (fp32 saturation sweep, fp64 "
                      "pressure solve) + fp32 matrix-core analysis, ensemble resident in HBM, correlated R (config 3)"}


def config4_sharded(device, comm, n_total=4096, n_grid=256):
    """BASELINE.json config 4: N_e = 4096 members at 256 x 256, members split over the ranks (4096 / world each: STRONG scaling --
    the total work is fixed), one ES-MDA pass = forward model of the local members (no communication) + the analysis step over
    the ranks (two all-reduces on the library's device buffers).  fp64 forward model, fp32 matrix-core analysis.  Called by EVERY rank."""
    from historymatching_amd.dist import es_mda_sharded, shard_bounds
    from historymatching_amd.geostat import gaussian_fields_kron

    lo, hi = shard_bounds(n_total, comm.world_size, comm.rank)
    model = build_model(64, device=device, n=n_grid)
    n_obs = NTIME * 4
    R12 = reference_obs_error(n_obs)
    obs = np.clip(0.2 + R12 @ np.random.RandomState(9).randn(n_obs), 0, 1)
    # every rank draws only its own members' fields (seeded per member block: the prior does not depend on the world size
    # up to the block boundaries; statistically the same prior -- this leg is a timing)
    prior = gaussian_fields_kron(n_grid, n_grid, 2, 1, hi - lo, r=0.8, seed=1000 + lo)
    st = {}
    comm.barrier()
    t0 = time.perf_counter()
    post = es_mda_sharded(model, prior, obs, R12, DT, NTIME, n_iter=1, seed=3, comm=comm, dtype=32, device=device, stats=st)
    comm.barrier()
    wall = comm.all_reduce_max(time.perf_counter() - t0)
    fwd_ms = comm.all_reduce_max(st["ms_forward"])
    upd_ms = comm.all_reduce_max(st["ms_update"] + st.get("ms_comm", 0.0))
    ok = comm.all_reduce_max(0.0 if np.isfinite(post).all() else 1.0) == 0.0
    return {"members_total": n_total, "members_per_rank": hi - lo, "grid": [n_grid, n_grid], "n_ranks": comm.world_size, "scaling": "strong",
            "passes": 1, "wall_s": wall, "device_ms_forward_max": fwd_ms, "device_ms_update_max": upd_ms,
            "member_steps_handed_to_the_cg": int(comm.all_reduce_max(float(st.get("nd_fallbacks", 0)))),  # (direct solver's a-posteriori check; max over ranks)
            "fallbacks": fallbacks_over_ranks(comm, st),
            "ensemble_steps_per_s_forward": n_total * NTIME / (fwd_ms * 1e-3), "ensemble_steps_per_s_incl_update_and_setup": n_total * NTIME / wall,
            "posterior_finite": bool(ok),
            "collective": "RCCL (hm_upd_run_comm)" if comm.rccl is not None else ("none (one rank)" if comm.world_size == 1 else f"host channel ({comm.rccl_error})")}


def config5_sharded(device, comm, n_total=1000, n_grid=512, bounded=False):
    """BASELINE.json config 5: N_e = 1000 members at 512 x 512, localised update (taper = bump(dist / 1.2) to the 4 producers x
    40 times, HistoryMatch.py:700-717, 863), fp32 plans, members split over the ranks (strong scaling); one ES-MDA pass = forward
    model of the local members + the localised analysis over the ranks: two all-reduces, per-element solves sharded by state
    column, all-gather of the weights.  Called by EVERY rank.  The whole ensemble from 4 ranks up (or with --config5: one rank
    ~2 min); below that `bounded`: ONE GPU's share of the 8-GPU configuration -- 125 members, the same grid, the same localised
    analysis step on those members -- so that the default line carries a driver-timed figure for this configuration."""
    from historymatching_amd.dist import es_mda_sharded, shard_bounds
    from historymatching_amd.geostat import gaussian_fields_kron
    from historymatching_amd.localization import taper_for_wells

    if bounded:
        n_total = 125
    lo, hi = shard_bounds(n_total, comm.world_size, comm.rank)
    model = build_model(32, device=device, n=n_grid)
    n_obs = NTIME * 4
    R12 = reference_obs_error(n_obs)
    obs = np.clip(0.2 + R12 @ np.random.RandomState(9).randn(n_obs), 0, 1)
    taper = taper_for_wells(model, model.xy2ind(*model.prd_xy.T), NTIME, radius=1.2)
    prior = gaussian_fields_kron(n_grid, n_grid, 2, 1, hi - lo, r=0.8, seed=2000 + lo)
    st = {}
    comm.barrier()
    t0 = time.perf_counter()
    post = es_mda_sharded(model, prior, obs, R12, DT, NTIME, n_iter=1, seed=3, comm=comm, dtype=32, taper=taper, device=device, stats=st)
    comm.barrier()
    wall = comm.all_reduce_max(time.perf_counter() - t0)
    fwd_ms = comm.all_reduce_max(st["ms_forward"])
    upd_ms = comm.all_reduce_max(st["ms_update"] + st.get("ms_comm", 0.0))
    ok = comm.all_reduce_max(0.0 if np.isfinite(post).all() else 1.0) == 0.0
    return {"members_total": n_total, "members_per_rank": hi - lo, "grid": [n_grid, n_grid], "n_ranks": comm.world_size, "scaling": "strong",
            "sample": ("one GPU's shard of the 8-GPU configuration: 125 of the 1000 members, the whole 512 x 512 grid, one localised ES-MDA pass"
                       if bounded else "the whole configuration"),
            "passes": 1, "wall_s": wall, "device_ms_forward_max": fwd_ms, "device_ms_localised_update_max": upd_ms,
            "member_steps_handed_to_the_cg": int(comm.all_reduce_max(float(st.get("nd_fallbacks", 0)))),
            "fallbacks": fallbacks_over_ranks(comm, st),
            "ensemble_steps_per_s_forward": n_total * NTIME / (fwd_ms * 1e-3), "posterior_finite": bool(ok), "dtype": "f32 saturation sweep + fp64 pressure, fp32 matrix-core analysis",
            "collective": "RCCL (hm_upd_run_comm: 2 all-reduces + all-gather of the column-sharded weights)" if comm.rccl is not None
                          else ("none (one rank)" if comm.world_size == 1 else f"host channel ({comm.rccl_error})")}


# Silent fall-backs of the forward kernels, reported with every leg that runs them (a run that took one is slower than it should be and
# says nothing else): member-steps handed to the CG by the direct pressure solver's a-posteriori check (256 x 256 / 512 x 512), member-steps
# of a workgroup-team sweep redone by the single-workgroup tiled sweep (a team gave up waiting for a neighbour), member-steps the float32
# slab sweep did twice because water reached a slab that sat the step out.
FALLBACK_KEYS = ("nd_fallbacks", "team_retries", "slab_redos")


def fallbacks_over_ranks(comm, st):
    """The counters of `st` (forward statistics of this rank), largest over the ranks."""
    return {k: int(comm.all_reduce_max(float(st.get(k, 0)))) for k in FALLBACK_KEYS}


def config2_strong_shard(device, n_shard=125, passes=3):
    """One rank's share of BASELINE config 2 split over 8 GPUs (STRONG scaling: N_e = 1000 / 8 = 125 members at 128 x 128, fp64): a 40-step
    forward pass of the shard, timed like the headline (inputs resident, best of `passes`).  A shard of fewer members than CUs runs the sweep as
    teams of two slab workgroups per member (sat128s) and levels 3..0 of the pressure solve one front per workgroup.  Reported BESIDE `value`,
    never as it; `implied_speedup_8_gpus` = 8 x this rate / the 1000-member rate of this run: no multi-GPU run stands behind it (the
    forward model has no communication between members, HistoryMatch.py:376-380)."""
    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.geostat import gaussian_fields_kron

    model = build_model(64, device=device)
    perms = gaussian_fields_kron(NX, NY, 2, 1, n_shard, r=0.8, seed=1)
    plan = ForwardPlan(model, n_shard, DT, NTIME, keep_history=True, device=device)
    best, st = 1e9, None
    for _ in range(passes + 1):
        plan.set_inputs(perms, None, transformed=False)
        plan.sync()
        t0 = time.perf_counter()
        plan.run(0, NTIME)
        st = plan.sync()
        best = min(best, time.perf_counter() - t0)
    _, prods, status = plan.outputs(want_wsats=False)
    plan.close()
    return {"members": n_shard, "value": n_shard * NTIME / best, "unit": "ensemble-steps/s", "ms_per_pass": 1e3 * best,
            "avg_launch_ms": {"pressure": st["ms_pressure"] / max(1, st["n_pressure_launches"]), "saturation": st["ms_saturation"] / max(1, st["n_saturation_launches"])},
            "fallbacks": {k: int(st.get(k, 0)) for k in FALLBACK_KEYS}, "status_ok": bool(not status.any() and np.isfinite(prods).all()),
            "what": "one rank's shard of config 2 over 8 GPUs (strong scaling), one GPU, one member block; not part of `value`"}


def load_profile_json(name):
    """Newest committed profiles/rNN/<name> (measured separately under rocprofv3 / from the built object), or None."""
    try:
        return json.loads(sorted((ROOT / "profiles").glob(f"r*/{name}"))[-1].read_text()), str(sorted((ROOT / "profiles").glob(f"r*/{name}"))[-1].relative_to(ROOT))
    except Exception:
        return None, None


def self_launch(n):
    """Start `n` ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / HM_AMD_RDZV* set), relay rank 0's stdout, return the
    largest exit code.  The parent imports nothing that touches the GPU."""
    import secrets
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdzv_dir = tempfile.mkdtemp(prefix="hm_amd_bench_")  # 0700, ours
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HM_AMD_RDZV=os.path.join(rdzv_dir, "rdzv"),
               HM_AMD_RDZV_KEY=secrets.token_hex(16), HSA_ENABLE_IPC_MODE_LEGACY="0", HM_BENCH_SELF_LAUNCHED="1")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].stdout.read().decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    try:
        os.rmdir(rdzv_dir)
    except OSError:
        pass
    if any(rcs):
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return max(abs(rc) for rc in rcs)


def dry_run(args):
    """The N-rank launch path without a device: rendezvous over the host channel, a barrier, a max-over-ranks time, one line."""
    from historymatching_amd.dist import Comm

    comm = Comm.from_env()
    comm.barrier()
    t0 = time.perf_counter()
    ranks = comm.host.all_gather((comm.rank, comm.local_rank, os.getpid()))
    comm.barrier()
    elapsed = comm.all_reduce_max(time.perf_counter() - t0)
    if comm.rank == 0:
        print(json.dumps({"metric": "ensemble-steps/sec", "value": None, "unit": "ensemble-steps/s", "n_gpus": comm.world_size, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "barrier_s": elapsed,
                          "ranks": {"world": comm.world_size, "seen": [list(r) for r in ranks], "self_launched": os.environ.get("HM_BENCH_SELF_LAUNCHED") == "1"}}), flush=True)
    comm.close()
    return 0 if comm.world_size == args.gpus else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=N_E, help="members per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-esmda", action="store_true", help="skip the 4-pass ES-MDA leg (config 3)")
    ap.add_argument("--no-host-call", action="store_true", help="skip the PCIe-inclusive leg (the drop-in call with host arrays in and out)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 leg (N_e=4096 at 256x256 over the ranks)")
    ap.add_argument("--no-two-streams", action="store_true", help="the timed region as ONE member block on one stream (profiling runs: one kernel shape per name, no overlapping launches); same as --blocks 1")
    ap.add_argument("--blocks", type=int, default=0, help="member blocks (HIP streams) of the timed region; 0 = the library's default for the ensemble (forward.default_blocks)")
    ap.add_argument("--config5", action="store_true", help="run the WHOLE config-5 leg (N_e=1000 at 512x512, localised) also below 4 ranks (default there: one GPU's shard of 125 members)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg")
    ap.add_argument("--no-strong-shard", action="store_true", help="skip the 125-member shard of config 2 (one rank's share of a strong-scaled run over 8 GPUs)")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--variant", type=int, default=0, help="0 = fastest kernels, 1 = generic kernels")
    ap.add_argument("--dry-run", action="store_true", help="no device work: the ranks rendezvous, agree on a time and rank 0 prints a line (tests of the N-rank launch path)")
    args = ap.parse_args()

    # `--gpus N` run plainly (no launcher set WORLD_SIZE): become the launcher.  N fresh child processes, one per GPU, BEFORE this
    # process has made any GPU call (a process that has initialised the GPU must never be replaced or forked); the parent only
    # waits and relays rank 0's line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    if args.dry_run:
        sys.exit(dry_run(args))

    from historymatching_amd import _lib
    from historymatching_amd.dist import Comm
    from historymatching_amd.forward import BlockedForwardPlan, ForwardPlan, default_blocks
    from historymatching_amd.geostat import gaussian_fields_kron

    comm = Comm.from_env()
    world, rank, local_rank = comm.world_size, comm.rank, comm.local_rank
    if os.environ.get("HM_BENCH_ALL_ON_DEVICE0") == "1":  # functional check of the N > 1 path on a one-GPU box: every rank on device 0
        local_rank = comm.local_rank = 0                   # (RCCL refuses that; the reductions then run over the host channel)
    if args.gpus != world:  # the line's n_gpus must be what was asked for: a mismatch is an error, not a silent one-GPU benchmark
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different rank count", file=sys.stderr)
        comm.close()
        sys.exit(4)

    # every rank needs a device of its own: each rank sees the same device count and takes the same decision, before any collective
    n_dev = int(_lib.load().hm_device_count())
    if world > n_dev and os.environ.get("HM_BENCH_ALL_ON_DEVICE0") != "1":
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} asks for {world} ranks with a GPU each but this node shows {n_dev} device(s): no line printed", file=sys.stderr)
        comm.close()
        sys.exit(4)
    ctx = _lib.Context.get(local_rank)
    if world > 1 or os.environ.get("HM_BENCH_FORCE_DIST") == "1":  # the env switch exercises the RCCL path with 1 rank
        comm.enable_rccl(ctx, force_single=True)
    # what the ranks saw: device count and device of every rank, whether the library's RCCL communicator spans all of them
    rank_info = comm.host.all_gather({"rank": rank, "local_rank": local_rank, "device": ctx.name(), "visible_devices": ctx.device_count(),
                                      "rccl": comm.rccl is not None}) if world > 1 else \
        [{"rank": 0, "local_rank": local_rank, "device": ctx.name(), "visible_devices": ctx.device_count(), "rccl": comm.rccl is not None}]
    model = build_model(64, device=local_rank)
    n_e = args.members
    perms = gaussian_fields_kron(NX, NY, 2, 1, n_e, r=0.8, seed=1 + rank)  # synthetic prior, SURVEY.md 8d
    # The timed region runs the ensemble the way the library's forward_model runs it (forward.make_forward_model): as
    # default_blocks(model, N) member blocks, each on a HIP stream of its own, launches interleaved time step by time step -- at
    # N_e = 1000 three blocks: the pressure solve of one block (latency at modest occupancy) runs beside the sweeps of the others
    # and fills the partial last rounds of their launches.  Members are independent; the results are bit-identical to one block's
    # (checked below).  Per-kernel launch times, which the roofline needs, are taken from a ONE-block pass of the same workload
    # right behind the timed region (kernels of different streams overlap in the blocked run: their event times are not
    # durations of one kernel on an otherwise idle device).
    n_blocks = 1 if (args.no_two_streams or args.variant != 0) else (args.blocks or default_blocks(model, n_e))
    plan = BlockedForwardPlan(model, n_e, DT, NTIME, keep_history=True, device=local_rank, blocks=n_blocks)
    plan.set_variant(args.variant, args.variant)
    plan.set_inputs(perms, None, transformed=False)  # inputs resident in HBM before the timed region

    for _ in range(args.warmup):
        plan.run(0, NTIME)
    plan.sync()

    comm.barrier()           # all ranks' devices are idle here: plan.sync() above is a stream synchronisation per block
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(0, NTIME)
    stats = plan.sync()      # hipStreamSynchronize on every block's launch stream + event read-out
    comm.barrier()
    elapsed = comm.all_reduce_max(time.perf_counter() - t0)

    _, prods, status = plan.outputs(want_wsats=False)
    ok = not status.any() and np.isfinite(prods).all()
    plan.close()
    timed_fallbacks = {k: int(stats.get(k, 0)) for k in FALLBACK_KEYS}  # (plans created for this run: the counters cover warm-up + timed region)
    ok = ok and timed_fallbacks["team_retries"] == 0  # a team sweep that fell back to the tiled sweep inside the timed region: not the number to report

    blocks_info = {"n": n_blocks, "bounds": plan.bounds, "fallbacks": timed_fallbacks, "device_ms_longest_block": {"total": stats["ms_total"], "pressure": stats["ms_pressure"], "saturation": stats["ms_saturation"]},
                   "how": "member blocks on HIP streams of their own, launches interleaved time step by time step from one host thread (forward.BlockedForwardPlan; "
                          "forward_model runs every large ensemble on the 128 x 128 kernels this way)"}
    if n_blocks > 1:
        # the same workload as ONE block on one stream: the per-kernel statistics below (and the roofline) come from here
        one = ForwardPlan(model, n_e, DT, NTIME, keep_history=True, device=local_rank)
        one.set_variant(args.variant, args.variant)
        one.set_inputs(perms, None, transformed=False)
        one.run(0, NTIME)
        one.sync()
        passes1 = max(1, min(args.steps, 3))
        t1 = time.perf_counter()
        for _ in range(passes1):
            one.run(0, NTIME)
        stats = one.sync()
        wall1 = (time.perf_counter() - t1) / passes1
        p1 = one.outputs(want_wsats=False)[1]
        one.close()
        blocks_info["one_block"] = {"value": n_e * NTIME / wall1, "unit": "ensemble-steps/s", "ms_per_pass": 1e3 * wall1, "passes": passes1,
                                    "producer_series_identical_to_the_blocked_run": bool(np.array_equal(p1, prods)),
                                    "note": "this rank's ensemble as one member block on one stream; `roofline`, `device_ms` and avg_launch_ms are this pass's"}
        ok = ok and blocks_info["one_block"]["producer_series_identical_to_the_blocked_run"]

    def guarded(leg):
        """Run a leg every rank takes part in; if it fails on any rank, every rank reports the error instead of a result
        (the legs after the timed region never cost the headline line)."""
        err, res = None, None
        try:
            res = leg()
        except Exception as e:
            err = f"rank {rank}: {type(e).__name__}: {e}"
        msgs = comm.host.all_gather(err) if world > 1 else [err]
        bad = [m for m in msgs if m]
        return {"error": "; ".join(bad)} if bad else res

    # The same workload through the drop-in call itself -- forward_model(perms) -> [wsats (N, 41, Nxy), prods] with host arrays in
    # and the whole 5.4 GB saturation history out (HistoryMatch.py:383-387).  PCIe-inclusive; reported beside `value`, never as it.
    host_call = None
    if world == 1 and args.variant == 0 and args.members == N_E and not args.no_host_call:
        try:
            from historymatching_amd.forward import make_forward_model
            fm = make_forward_model(model, DT, NTIME, return_history=True)
            fm(perms)  # builds and keeps the device plan
            best = 1e9
            for _ in range(2):
                t1 = time.perf_counter()
                w_h, p_h = fm(perms)
                best = min(best, time.perf_counter() - t1)
            host_call = {"value": n_e * NTIME / best, "unit": "ensemble-steps/s", "wall_s": best, "device_s": model.last_stats["ms_total"] / 1e3,
                         "bytes_in": int(perms.nbytes), "bytes_out": int(w_h.nbytes + p_h.nbytes),
                         "producer_series_identical_to_device_resident_run": bool(np.array_equal(p_h, prods)),
                         "how": "forward_model(perms) of the host mirror: host arrays in, saturation history of every member and step out "
                                "(copied out time index by time index while the run goes on: hm_fwd_run_to_host); as forward_model runs every large ensemble "
                                "on these kernels: member blocks on streams of their own (forward.default_blocks), a host thread each"}
            fm.release()
            del w_h, p_h, fm
        except Exception as e:
            host_call = {"error": str(e)}

    def report(upd_sharded, c4, c5):
        """Rank 0: the one JSON line, from the headline's statistics and whatever the later legs produced."""
        member_steps = n_e * NTIME * args.steps * world
        value = member_steps / elapsed
        # ---- roofline of the dominant kernel (HIP events recorded on the launch stream, inside the library)
        w = 8
        nxy = NX * NY
        nts = stats["mean_nts"]
        sat_ms = stats["ms_saturation"] / max(1, stats["n_saturation_launches"])
        prs_ms = stats["ms_pressure"] / max(1, stats["n_pressure_launches"])
        dominant = "saturation" if stats["ms_saturation"] >= stats["ms_pressure"] else "pressure"
        isa, isa_src = load_profile_json("isa_counts.json")
        pmc, pmc_src = load_profile_json("pmc_hbm_traffic.json")
        # the committed counts (DP instructions per cell and sub-step, executed / algorithmic ratio) were taken from particular builds of
        # sat128r.o / press_nd.o: if the objects this process runs differ, `stale_inputs` says so (a separate flag -- `frac` is always this
        # run's launch time priced with those counts; the defaults below are the round-3 census)
        sys.path.insert(0, str(ROOT / "profiles" / "tools"))
        from obj_hash import object_hashes
        built = object_hashes()
        # saturation sweep (k_sat128r): member state register/LDS resident, bound by the CU's double-precision VALU.  Work per
        # launch = DP VALU instructions of the sub-step loop (counted from the built object: profiles/tools/isa_count.py) x
        # cells x sub-steps x members, in lane-instructions; the peak is one DP lane-instruction per lane-slot.
        dp_per_cell = (isa or {}).get("k_sat128r", {}).get("dp_valu_per_cell_substep", 21.0)
        # dry bands are skipped (sat128r.hip): the instructions actually executed are fewer than cells x sub-steps x count; the
        # ratio is measured (SQ_INSTS_VALU over the same workload, profiles/rNN/fp64_roofline.json) -- `frac` uses EXECUTED work
        f64r, f64r_src = load_profile_json("fp64_roofline.json")
        executed_ratio = ((f64r or {}).get("kernels", {}).get("k_sat128r", {}).get("executed_over_algorithmic", 1.0)) if args.variant == 0 else 1.0
        recorded = {"isa_counts.json": (isa or {}).get("object_sha256"), "fp64_roofline.json": (f64r or {}).get("object_sha256")}
        stale = [f"{src}: {obj}" for src, h in recorded.items() for obj in ("sat128r.o", "press_nd.o")
                 if not h or h.get(obj) != built.get(obj)] if args.variant == 0 else []
        sat_lane_instr_algorithmic = dp_per_cell * nxy * nts * n_e
        sat_lane_instr = sat_lane_instr_algorithmic * executed_ratio
        # pressure (press_nd.hip, nested dissection): fp64 matrix cores; flops = the v_mfma_f64_16x16x4 instructions the factorisation
        # issues per member (counted from the symbolic tables the kernels read, hm_debug_nd_tables) x 2 x 16 x 16 x 4
        prs_mfma_per_member = nd_mfma_count() if args.variant == 0 else 36 * 4 * 8 * NX
        prs_flops = 2048.0 * prs_mfma_per_member * n_e
        if dominant == "saturation":
            ach = 2 * sat_lane_instr / (sat_ms * 1e-3) / 1e12
            bound, what = "fp64_valu", ("double-precision VALU issue slots actually executed: every DP VALU instruction of the sub-step loop counted as "
                                        "one FMA slot (2 flop) per lane, dry bands' skipped instructions not counted; peak = 256 CUs x 4 SIMDs x 16 DP lanes x 2.4 GHz x 2")
        else:
            ach = prs_flops / (prs_ms * 1e-3) / 1e12
            bound, what = "fp64_mfma", "v_mfma_f64_16x16x4 flops of the nested-dissection factorisation (tile products of the fronts)"
        traffic = None
        key = {"saturation": "sat128", "pressure": "press_nd"}[dominant]
        if pmc and args.variant == 0 and key in pmc.get("kernels", {}):
            traffic = pmc["kernels"][key]["hbm_bytes_per_member_corrected"] * n_e
        dom_ms = sat_ms if dominant == "saturation" else prs_ms
        # SURVEY.md 8d accounting kept as a separately named diagnostic: "effective" bytes (read S,Vx,Vy + write S per explicit
        # sub-step; compulsory + factor write + factor read for the pressure solver) -- NOT a roofline for a register-resident sweep
        sat_bytes = w * nxy * 4 * nts * n_e
        prs_bytes = w * (4 * nxy + 2 * ND_FACTOR_DOUBLES + 2 * ND_ARENA_DOUBLES) * n_e  # compulsory + factor write/read + update matrices write/read
        roofline = {
            "bound": bound, "kernel": {"saturation": "k_sat128r", "pressure": "k_nd_* (press_nd.hip: assemble, sub, wave x3, top, solve)"}[dominant] if args.variant == 0 else dominant,
            "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
            # `frac` prices THIS run's launch time with the committed instruction counts (ISA census, executed / algorithmic ratio from a
            # counter pass): `frac_valid` says whether those counts were taken from the very objects that ran (sha256); when they were not
            # (`stale_inputs`), `frac` is a number priced with another build's counts -- kept, but flagged.  The executed / algorithmic
            # ratio is a committed measurement (profiles/rNN/fp64_roofline.json), not re-measured in this run.
            "frac_valid": not stale, "executed_ratio_measured_in_this_run": False,
            "launch_times_measured_in": ("the timed region" if n_blocks == 1 else
                                         f"a one-block pass of the same workload right behind the timed region ({blocks_info['one_block']['passes']} passes, HIP events on the "
                                         "launch stream): the timed region runs the ensemble as member blocks on several streams, whose kernels overlap"),
            "stale_inputs": bool(stale), "stale_inputs_detail": stale or None,
            "inputs_taken_from_objects": recorded, "objects_run": {k: built.get(k) for k in ("sat128r.o", "press_nd.o")},
            "unit_note": "for bound fp64_valu `achieved` is an issue-slot rate: DP lane-instructions/s x 2 (every DP VALU instruction priced as one FMA "
                         "slot, whether it is an FMA, an add, a compare or part of a division) against 2 x the lane-slot peak; the plain rate is "
                         "in achieved_dp_lane_instr_per_s / peak_dp_lane_instr_per_s",
            "achieved_dp_lane_instr_per_s": sat_lane_instr / (sat_ms * 1e-3) if dominant == "saturation" else None,
            "peak_dp_lane_instr_per_s": DP_LANE_RATE,
            "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None,
            "hbm_frac_of_peak_from_measured_traffic": None if traffic is None else traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "work_counted": what, "dp_valu_per_cell_substep": dp_per_cell, "isa_count_source": isa_src,
            "executed_over_algorithmic_instructions": executed_ratio, "executed_ratio_source": f64r_src,
            "frac_if_skipped_dry_bands_counted_as_work": 2 * sat_lane_instr_algorithmic / (sat_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            "measured_issue_ceiling": {"frac_of_peak": 0.74, "note": "at this kernel's two waves per SIMD a SIMD issues one simple DP instruction per "
                                       "5.4 cycles, not 4 (v_rcp_f64: 16.4; an FMA with three distinct register operands: 6.9) -- "
                                       "profiles/diag/valu_rate.hip, profiles/README.md"},
            "avg_launch_ms": {"saturation": sat_ms, "pressure": prs_ms},
            "per_kernel": {"saturation_fp64_valu_frac": 2 * sat_lane_instr / (sat_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                           # the factorisation's matrix instructions counted from the symbolic tables, every front eliminated: since fronts
                           # whose subtree is still dry are skipped (k_nd_plan) the EXECUTED work is lower -- the counter-based figure below
                           "pressure_fp64_mfma_frac_skip_free_count": prs_flops / (prs_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                           # the same solve by round 2's block elimination takes 36 x 4 x 8 x Nx matrix instructions per member (302 Mflop
                           # against the nested dissection's 55): this launch time priced at THAT flop count, for comparison across rounds only
                           "pressure_frac_at_block_elimination_flop_count": 2048.0 * 36 * 4 * 8 * NX * n_e / (prs_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                           # counter-based: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop per launch (committed PMC pass) over THIS run's launch time
                           "pressure_fp64_mfma_frac_from_counters": None if not (f64r or {}).get("pressure_nd") else
                           f64r["pressure_nd"]["fp64_mfma_flops_per_member_step"] * n_e / (prs_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
            "effective_bandwidth_diagnostic": {
                "note": "SURVEY.md 8d 'effective GB/s' (algorithmic bytes / launch time); exceeds the HBM peak by construction for the register-resident sweep",
                "saturation_GBps": sat_bytes / (sat_ms * 1e-3) / 1e9, "pressure_GBps": prs_bytes / (prs_ms * 1e-3) / 1e9,
                "algorithmic_bytes_per_launch": {"saturation": sat_bytes, "pressure": prs_bytes},
                "compulsory_floor_bytes_per_member_step": 4 * w * nxy},
            "pressure": {
                "kernel": "k_nd_* (press_nd.hip)" if args.variant == 0 else "pressure",
                "avg_launch_ms": prs_ms,
                # both bounds of the pressure step: matrix flops EXECUTED (counters: fronts still dry are skipped) over the fp64 matrix peak,
                # and measured HBM traffic over the HBM peak -- it is bound by neither (latency at low occupancy, profiles/README.md)
                "fp64_mfma_frac": None if not (f64r or {}).get("pressure_nd") else
                f64r["pressure_nd"]["fp64_mfma_flops_per_member_step"] * n_e / (prs_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "fp64_mfma_frac_skip_free_count": prs_flops / (prs_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "hbm_traffic_bytes_per_launch": None if not (pmc and "press_nd" in pmc.get("kernels", {})) else pmc["kernels"]["press_nd"]["hbm_bytes_per_member_corrected"] * n_e,
                "hbm_frac": None if not (pmc and "press_nd" in pmc.get("kernels", {})) else
                pmc["kernels"]["press_nd"]["hbm_bytes_per_member_corrected"] * n_e / (prs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "sources": [f64r_src, pmc_src]},
            "mean_nts": nts, "assumes": "2.4 GHz, 256 CUs (the chip holds a lower clock under this load: profiles/rNN/fp64_roofline.json)",
        }
        upd = None if upd_sharded is None else {"sharded": upd_sharded}
        if world == 1:
            try:
                upd = dict(upd or {}, **es_update_timing(local_rank))
            except Exception as e:  # the forward metric stands on its own
                upd = dict(upd or {}, error=str(e))
            if not args.no_esmda:
                try:
                    upd = dict(upd or {}, es_mda_config3=es_mda_c3(local_rank, perms))
                    c3 = upd["es_mda_config3"]
                    # the HEADLINE analysis figure: the step where it is used -- inside the 4-pass assimilation, each step behind a forward pass
                    # of 40 time steps (cold caches, whatever clocks the device holds there); the back-to-back figure above is the secondary
                    in_situ = c3["device_ms_update"] / c3["iterations"]
                    flops = 4.0 * N_E * 160 * NX * NY
                    upd["in_situ_ms"] = in_situ
                    upd["mfma_frac_in_situ"] = flops / (in_situ * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
                    upd["headline"] = "in_situ_ms / mfma_frac_in_situ (one analysis step inside es_mda_config3); wall_ms / mfma_frac_of_fp32_peak are the same step queued back to back"
                except Exception as e:
                    upd = dict(upd or {}, es_mda_config3={"error": str(e)})
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            nproc = os.cpu_count() or 1
            members = max(nproc, 8)
            v, wall = cpu_baseline(members, args.cpu_steps, nproc)
            cpu = {"value": v, "unit": "ensemble-steps/s", "cores": nproc, "kind": "port",
                   "sample": f"{members} members x {args.cpu_steps} steps of the same 128x128 workload, "
                             f"{nproc} processes x 1 BLAS thread, {wall:.1f} s wall"}
            try:
                cpu["config1"] = config1_reference_run(local_rank, nproc)
            except Exception as e:
                cpu["config1"] = {"error": f"{type(e).__name__}: {e}"}
        strong = None
        if world == 1 and args.members == N_E and args.variant == 0 and not args.no_strong_shard:
            try:
                strong = config2_strong_shard(local_rank)
                strong["implied_speedup_8_gpus"] = 8.0 * strong["value"] / value
                strong["implied_speedup_note"] = ("8 x the 125-member shard's rate / this run's 1000-member rate: what splitting config 2's N_e = 1000 over 8 GPUs "
                                                  "would give with no communication; a projection from one GPU, not a measured scaling curve")
            except Exception as e:
                strong = {"error": f"{type(e).__name__}: {e}"}
        out = {
            "metric": "ensemble-steps/sec", "value": value, "unit": "ensemble-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # (near the front so that a truncated tail of the line keeps them: the analysis step where it is used, and the silent fall-backs)
            "es_update_headline": None if not upd else {k: upd.get(k) for k in ("in_situ_ms", "mfma_frac_in_situ", "wall_ms", "mfma_frac_of_fp32_peak")},
            "fallbacks_in_timed_region": timed_fallbacks,
            "config": {"workload": f"N_e={n_e} per GPU, {NX}x{NY} grid, forward model only (nTime={NTIME}, dt={DT}), fp64",
                       "members_per_gpu": n_e, "grid": [NX, NY], "nTime": NTIME, "device": ctx.name(),
                       "kernel_variant": args.variant, "results_finite_and_status_ok": bool(ok),
                       "ranks": "one process per GPU, host channel for barriers/timing, RCCL from the library for the update's reductions (no PyTorch)"},
            "ranks": {"world": world, "rccl_communicator_ranks": (world if all(r["rccl"] for r in rank_info) else 0), "rccl_error": comm.rccl_error,
                      "per_rank": rank_info, "self_launched": os.environ.get("HM_BENCH_SELF_LAUNCHED") == "1"},
            "roofline": roofline, "cpu_baseline": cpu, "blocks": blocks_info, "host_call": host_call, "es_update": upd, "config4": c4, "config5": c5,
            "config2_strong_shard": strong,
            "device_ms": {"total": stats["ms_total"], "pressure": stats["ms_pressure"], "saturation": stats["ms_saturation"]},
        }
        print(json.dumps(out), flush=True)

    # The legs below run collectives between the ranks.  Should one of them hang, the headline -- measured and complete at this
    # point -- is still reported: a watchdog prints the line without them and ends the process.
    legs_done = threading.Event()
    deadline = float(os.environ.get("HM_BENCH_LEG_DEADLINE", "600"))

    def watchdog():
        if legs_done.wait(deadline):
            return
        late = {"error": f"did not finish within {deadline:.0f} s (HM_BENCH_LEG_DEADLINE); the line was printed by the watchdog"}
        if rank == 0:
            report(late, late, late)
        os._exit(0 if ok else 3)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()
    upd_sharded = c4 = None
    if world > 1 or comm.rccl is not None:  # every rank: the analysis step sharded over the ranks (outside the timed region of `value`)
        upd_sharded = guarded(lambda: es_update_sharded_timing(local_rank, comm))
    if not args.no_config4 and args.members == N_E:
        c4 = guarded(lambda: config4_sharded(local_rank, comm))
    c5 = None
    if args.members == N_E and not args.no_config5:
        whole = args.config5 or world >= 4
        c5 = guarded(lambda: config5_sharded(local_rank, comm, bounded=not whole))

    legs_done.set()
    if rank == 0:
        report(upd_sharded, c4, c5)
    comm.barrier()
    comm.close()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
