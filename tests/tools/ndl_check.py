"""Nested-dissection pressure solve of the larger grids (press_nd256.o / press_nd512.o: the big-front kernels) against the two-level CG
(press_variant 15) and the oracle: pressures and fluxes of a few members on a part-swept saturation field, then the launch time at N members.

    python tests/tools/ndl_check.py [n=256 [N=512 [oracle=1]]]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from oracle.ressim import perm_transf, set_perm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
with_oracle = int(sys.argv[3]) if len(sys.argv) > 3 else 1
om, gm = make_models(n, n)
M = 3
x = perms(n, n, M, seed=3)
rng = np.random.RandomState(0)
S = np.clip(0.3 * rng.rand(M, n * n) * (rng.rand(M, n * n) < 0.3), 0, 1)
res = {}
for v in (15, 0):
    plan = ForwardPlan(gm, M, 0.025, 1, keep_history=True, device=0)
    plan.set_variant(v, 0)
    plan.set_inputs(perm_transf(x), None, transformed=True)
    plan.set_field("S", S)
    plan.pressure_only(0)
    st = plan.sync()
    res[v] = {k: plan.get_field(k).copy() for k in ("P", "Vx", "Vy", "TX", "TY")}
    _, _, status = plan.outputs()
    print(f"variant {v}: status {status} mean_n_cg {st['mean_n_cg']}", flush=True)
    plan.close()
assert np.array_equal(res[15]["TX"], res[0]["TX"]) and np.array_equal(res[15]["TY"], res[0]["TY"]), "assembly differs"
q = om.source_field(0)[0]
for m in range(M):
    P0, P15 = res[0]["P"][m].ravel(), res[15]["P"][m].ravel()
    Vx, Vy = res[0]["Vx"][m], res[0]["Vy"][m]
    div = (Vx[1:] - Vx[:-1]) + (Vy[:, 1:] - Vy[:, :-1])
    print(f"member {m}: max|P_nd - P_cg| / max|P| = {np.abs(P0 - P15).max() / np.abs(P15).max():.2e}   max|V_nd - V_cg| = "
          f"{max(np.abs(res[0]['Vx'][m] - res[15]['Vx'][m]).max(), np.abs(res[0]['Vy'][m] - res[15]['Vy'][m]).max()):.2e}   "
          f"max|div V - q| = {np.abs(div.ravel() - q).max():.2e}", flush=True)
    if with_oracle:
        set_perm(om, x[m])
        Po, Vxo, Vyo = om.pressure_step(S[m], q)
        for v in (15, 0):
            P, Vxv, Vyv = res[v]["P"][m].ravel(), res[v]["Vx"][m].ravel(), res[v]["Vy"][m].ravel()
            print(f"   variant {v:2d}: max|P - P_oracle| / max|P| = {np.abs(P - Po.ravel()).max() / np.abs(Po).max():.2e}   "
                  f"max|V - V_oracle| = {max(np.abs(Vxv - Vxo.ravel()).max(), np.abs(Vyv - Vyo.ravel()).max()):.2e}", flush=True)

if N > 0:
    xN = perms(n, n, N, seed=1)
    for v in (0, 15):
        plan = ForwardPlan(gm, N, 0.025, 3, keep_history=False, device=0)
        plan.set_variant(v, 0)
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        plan.sync()
        plan.set_inputs(xN, None, transformed=False)
        plan.run()
        st = plan.sync()
        _, _, status = plan.outputs(want_wsats=False)
        print(f"variant {v:2d}: pressure {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms/launch, saturation {st['ms_saturation'] / st['n_saturation_launches']:.2f} ms/launch "
              f"over 3 time steps, {N} members; status ok {not status.any()}", flush=True)
        plan.close()
