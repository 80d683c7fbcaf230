"""Nested-dissection pressure solve (press_variant 12; 14 = every front eliminated every step) against the block elimination (variant 13)
and the oracle: pressures and fluxes of a few members on a part-swept saturation field, then the launch averages over a whole run at N members.

    python tests/tools/nd_check.py [N=1000]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from oracle.ressim import perm_transf, set_perm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 128
om, gm = make_models(n, n)
M = 4
x = perms(n, n, M, seed=3)
rng = np.random.RandomState(0)
S = np.clip(0.3 * rng.rand(M, n * n) * (rng.rand(M, n * n) < 0.3), 0, 1)
res = {}
for v in (13, 12):
    plan = ForwardPlan(gm, M, 0.025, 1, keep_history=True, device=0)
    plan.set_variant(v, 0)
    plan.set_inputs(perm_transf(x), None, transformed=True)
    plan.set_field("S", S)
    plan.pressure_only(0)
    plan.sync()
    res[v] = {k: plan.get_field(k).copy() for k in ("P", "Vx", "Vy", "TX", "TY")}
    _, _, status = plan.outputs()
    print(f"variant {v}: status {status}", flush=True)
    plan.close()
for m in range(M):
    set_perm(om, x[m])
    Po, Vxo, Vyo = om.pressure_step(S[m], om.source_field(0)[0])
    for v in (13, 12):
        P, Vx, Vy = res[v]["P"][m].ravel(), res[v]["Vx"][m].ravel(), res[v]["Vy"][m].ravel()
        print(f"member {m} variant {v:2d}: max|P - P_oracle| / max|P| = {np.abs(P - Po.ravel()).max() / np.abs(Po).max():.2e}   "
              f"max|V - V_oracle| = {max(np.abs(Vx - Vxo.ravel()).max(), np.abs(Vy - Vyo.ravel()).max()):.2e}", flush=True)
assert np.array_equal(res[13]["TX"], res[12]["TX"]) and np.array_equal(res[13]["TY"], res[12]["TY"])

if N > 0:  # launch averages over a whole run (the nested dissection skips fronts that are still dry: a solve on the initial state says nothing)
    xN = perms(n, n, N, seed=1)
    for v in (13, 14, 12):
        plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False, device=0)
        plan.set_variant(v, 0)
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        plan.sync()
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        st = plan.sync()
        print(f"variant {v:2d}: {st['ms_pressure'] / st['n_pressure_launches']:.3f} ms/launch (events) averaged over a run of 40 time steps, {N} members", flush=True)
        plan.close()
