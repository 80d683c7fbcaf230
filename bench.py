#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ensemble forward model.

Metric (BASELINE.json): ensemble-steps/sec = N_e * nTime / wall(forward_model), workload = config[1]:
N_e=1000 members, 128x128 grid, forward model only (TPFA pressure solve + explicit upwind saturation sweep per
member per step), fp64, nTime=40, wells/fluid/dt of notebooks/HistoryMatch.py:97,177-190,219-221.

A "step" (--steps) is ONE pass of the hot path over the whole batch: all N_e members advanced nTime=40 time
steps, inputs (permeability, initial saturation) already resident in HBM.  Multi-GPU (--gpus N, launched by
torch.distributed.run): members are independent, so every rank runs its own N_e members with no data-path
collective ("weak" scaling); value = total member-steps of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library on the launch
stream) and `cpu_baseline` (the NumPy/SciPy oracle on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NX = NY = 128
N_E = 1000
NTIME = 40
DT = 0.025
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def build_model(dtype=64, device=None):
    from historymatching_amd.ressim import ResSim

    m = ResSim(NX, NY, 2, 1, dtype=dtype, device=device)
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


def es_update_timing(device):
    """BASELINE.json's second metric (ES-MDA update wall-time) at config 3's shape: N=1000 members, M=128*128 state
    elements, n_obs=160, fp32 state contractions on the matrix cores, every N x n_obs quantity in fp64.  Device time of
    one analysis step (HIP events inside the library), inputs resident in HBM; flops by SURVEY.md 8d (min-flop order)."""
    from historymatching_amd.update import UpdatePlan

    N, M, n_obs = N_E, NX * NY, 160
    rng = np.random.RandomState(0)
    plan = UpdatePlan(N, N, M, n_obs, dtype=32, device=device)
    plan.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), rng.rand(n_obs), 0.1 * rng.randn(N, n_obs), 3.0 * np.eye(n_obs))
    plan.run_local()  # warm-up
    ms = sorted(plan.run_local()["ms_update"] for _ in range(7))
    plan.close()
    flops = 4.0 * N * n_obs * M
    peak = 157.3  # TF, fp32 matrix peak (MI355X_MICROARCH.md)
    return {"wall_ms": ms[len(ms) // 2], "best_ms": ms[0], "config": f"N={N}, M={M}, n_obs={n_obs}, fp32 (config 3 shape)",
            "flops_min_order": flops, "tflops": flops / (ms[len(ms) // 2] * 1e-3) / 1e12, "mfma_peak_tflops": peak,
            "mfma_frac_of_fp32_peak": flops / (ms[len(ms) // 2] * 1e-3) / 1e12 / peak}


def es_update_sharded_timing(device, world, reps=5):
    """The analysis step row-sharded over the ranks (weak scaling like the forward metric: every rank holds
    N_e=1000 members of an N_e x world ensemble, M=128*128, n_obs=160, fp32 state): three local phases with two RCCL
    all-reduces on the library's own device buffers in between (column sums; X^T S and S^T S).  Wall time per update,
    barrier + device synchronisation on both sides, maximum over ranks.  Called by EVERY rank."""
    import torch
    import torch.distributed as td

    from historymatching_amd.dist import Comm, sharded_update
    from historymatching_amd.update import UpdatePlan

    N, M, n_obs = N_E, NX * NY, 160
    rng = np.random.RandomState(100 + td.get_rank())
    plan = UpdatePlan(N * world, N, M, n_obs, dtype=32, device=device)
    plan.set_inputs(rng.randn(N, M), rng.rand(N, n_obs), np.random.RandomState(7).rand(n_obs), 0.1 * rng.randn(N, n_obs), 3.0 * np.eye(n_obs))
    comm = Comm()
    sharded_update(plan, comm, fetch=False)  # warm-up (RCCL communicator set-up included)
    td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sharded_update(plan, comm, fetch=False)
    td.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device="cuda")
    td.all_reduce(t, op=td.ReduceOp.MAX)
    nbytes = sum(plan.reduce_buffer(w)[1] * np.dtype(plan.reduce_buffer(w)[2]).itemsize for w in (0, 1, 2, 3))
    plan.close()
    return {"wall_ms": 1e3 * float(t.item()), "n_ranks": world, "members_total": N * world, "members_per_rank": N, "M": M, "n_obs": n_obs,
            "allreduce_bytes_per_update": int(nbytes), "collective": "RCCL all-reduce in place on the library's device buffers",
            "config": "row-sharded global analysis step, fp32 state contractions on the matrix cores, weak scaling"}


def es_mda_c3(device, perms, n_iter=4):
    """BASELINE.json config 3: N_e=1000, 128x128, 4 ES-MDA passes (forward model in fp32 mode + fp32 matrix-core analysis), the
    ensemble resident in HBM throughout (update.es_mda_device).  Observations = member 0's simulated production + noise."""
    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.update import es_mda_device

    model = build_model(32, device=device)  # config 3 is the fp32 configuration: fp32 saturation sweep (sat128f), fp64 pressure
    n_obs = NTIME * 4
    rng = np.random.RandomState(4)
    R12 = 0.1 * np.eye(n_obs)  # HistoryMatch.py:243-259 uses a correlated R; the update cost does not depend on it
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
    return {"iterations": n_iter, "wall_s": wall, "device_ms_forward": st["ms_forward"], "device_ms_update": st["ms_update"],
            "ensemble_steps_per_s_incl_updates": len(perms) * NTIME * n_iter / wall,
            "posterior_finite": bool(np.isfinite(post).all()),
            "config": f"N_e={len(perms)}, {NX}x{NY}, {n_iter} ES-MDA passes, forward model dtype=32 (fp32 saturation sweep, fp64 "
                      "pressure solve) + fp32 matrix-core analysis, ensemble resident in HBM (config 3)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=N_E, help="members per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-esmda", action="store_true", help="skip the 4-pass ES-MDA leg (config 3)")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--variant", type=int, default=0, help="0 = fastest kernels, 1 = generic kernels")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    td = None
    if world > 1 or os.environ.get("HM_BENCH_FORCE_DIST") == "1":  # the env switch exercises the N>1 code path with 1 rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as td

        torch.cuda.set_device(local_rank)
        td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from historymatching_amd import _lib
    from historymatching_amd.forward import ForwardPlan
    from historymatching_amd.geostat import gaussian_fields_kron

    ctx = _lib.Context.get(local_rank)
    model = build_model(64, device=local_rank)
    n_e = args.members
    perms = gaussian_fields_kron(NX, NY, 2, 1, n_e, r=0.8, seed=1 + rank)  # synthetic prior, SURVEY.md 8d
    plan = ForwardPlan(model, n_e, DT, NTIME, keep_history=True, device=local_rank)
    plan.set_variant(args.variant, args.variant)
    plan.set_inputs(perms, None, transformed=False)  # inputs resident in HBM before the timed region

    def one_pass():
        plan.run(0, NTIME)

    for _ in range(args.warmup):
        one_pass()
    plan.sync()

    def fence():
        if td is not None:
            import torch

            td.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    stats = plan.sync()  # hipStreamSynchronize on the launch stream + event read-out
    fence()
    elapsed = time.perf_counter() - t0
    if td is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    _, prods, status = plan.outputs(want_wsats=False)
    ok = not status.any() and np.isfinite(prods).all()

    upd_sharded = None
    if td is not None:  # every rank: the analysis step sharded over the ranks (outside the timed region of `value`)
        try:
            upd_sharded = es_update_sharded_timing(local_rank, world)
        except Exception as e:
            upd_sharded = {"error": str(e)}

    if rank == 0:
        member_steps = n_e * NTIME * args.steps * world
        value = member_steps / elapsed
        # ---- roofline of the dominant kernel (HIP events recorded on the launch stream, inside the library)
        w = 8
        nxy = NX * NY
        nts = stats["mean_nts"]
        sat_ms = stats["ms_saturation"] / max(1, stats["n_saturation_launches"])
        prs_ms = stats["ms_pressure"] / max(1, stats["n_pressure_launches"])
        # algorithmic bytes per member-step: saturation = read S,Vx,Vy + write S per explicit
        # sub-step; pressure = compulsory (K,S in; P out) + factor write + factor read of the direct block solver
        # (the symmetric factor: 36 of the 64 16x16 tiles of every 128x128 inverse Schur complement)
        sat_bytes = w * nxy * 4 * nts * n_e
        prs_bytes = w * nxy * (4 + 2 * NY * 36 / 64) * n_e
        dominant = "saturation" if stats["ms_saturation"] >= stats["ms_pressure"] else "pressure"
        ach = (sat_bytes / (sat_ms * 1e-3) if dominant == "saturation" else prs_bytes / (prs_ms * 1e-3)) / 1e9
        # measured HBM traffic of the same kernel: PMC passes are taken separately under rocprofv3 (profiles/),
        # bench.py only scales the per-member figure to this launch size
        traffic, traffic_src = None, None
        try:
            pmc_file = sorted((ROOT / "profiles").glob("r*/pmc_hbm_traffic.json"))[-1]
            pmc = json.loads(pmc_file.read_text())
            key = {"saturation": "sat128", "pressure": "press128s"}[dominant]
            if args.variant == 0 and key in pmc["kernels"]:
                traffic = pmc["kernels"][key]["hbm_bytes_per_member_corrected"] * n_e
                traffic_src = str(pmc_file.relative_to(ROOT))
        except Exception:
            pass
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": {"saturation": sat_ms, "pressure": prs_ms},
            "algorithmic_bytes_per_launch": {"saturation": sat_bytes, "pressure": prs_bytes},
            "compulsory_floor_bytes_per_member_step": 4 * w * nxy,
            "mean_nts": nts,
            "note": "effective GB/s by SURVEY.md 8d accounting; the state is LDS/register-resident so it may exceed HBM peak",
            # what actually bounds the two kernels: the CU's fp64 pipe.  Saturation: 37 double-precision
            # VALU instructions per cell and explicit sub-step (sat128.hip ISA), peak = CUs x 4 SIMDs x 16 lanes x clock
            "compute_view": {
                "saturation_fp64_valu_frac": (37.0 * nxy * nts * n_e / (sat_ms * 1e-3)) / (256 * 4 * 16 * 2.4e9),
                "pressure_fp64_mfma_frac": (2.0 * 36 * 16 * 16 * 16 * 8 * NX * n_e / (prs_ms * 1e-3)) / 78.6e12,
                "assumes": "2.4 GHz, 256 CUs; pressure flops = rank-16 updates of the 36 stored tiles x 8 panels x Nx blocks",
            },
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
        out = {
            "metric": "ensemble-steps/sec", "value": value, "unit": "ensemble-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N_e={n_e} per GPU, {NX}x{NY} grid, forward model only (nTime={NTIME}, dt={DT}), fp64",
                       "members_per_gpu": n_e, "grid": [NX, NY], "nTime": NTIME, "device": ctx.name(),
                       "kernel_variant": args.variant, "results_finite_and_status_ok": bool(ok)},
            "roofline": roofline, "cpu_baseline": cpu, "es_update": upd,
            "device_ms": {"total": stats["ms_total"], "pressure": stats["ms_pressure"], "saturation": stats["ms_saturation"]},
        }
        print(json.dumps(out))
    plan.close()
    if td is not None:
        td.barrier()
        td.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
