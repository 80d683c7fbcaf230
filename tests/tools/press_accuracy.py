"""Accuracy of the pressure variants against a long-double-refined solution of the same TPFA system (one member)."""
import sys
from pathlib import Path

import numpy as np
import scipy.sparse.linalg as sla

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle.ressim as R  # noqa: E402
from helpers import make_models, perms  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from oracle.ressim import perm_transf, set_perm  # noqa: E402

n = 128
om, gm = make_models(n, n)
x = perms(n, n, 1, seed=3)
set_perm(om, x[0])
S = np.full(n * n, 0.15)
q = om.source_field(0)[0]
H = {}
orig = R.spsolve
R.spsolve = lambda A, b: (H.update(A=A.tocsc(), b=b), sla.spsolve(A.tocsc(), b))[1]
Po, Vxo, Vyo = om.pressure_step(S, q)
R.spsolve = orig
A, b = H["A"], H["b"]
lu = sla.splu(A)
xr = lu.solve(b).astype(np.longdouble)
Al = A.astype(np.float64)
for _ in range(5):  # iterative refinement with long-double residuals
    r = b.astype(np.longdouble) - (Al @ xr.astype(np.float64)).astype(np.longdouble)
    # residual in extended precision: accumulate per row
    Ac = A.tocsr()
    rr = np.array([b[i] - np.dot(Ac.data[Ac.indptr[i]:Ac.indptr[i + 1]].astype(np.longdouble), xr[Ac.indices[Ac.indptr[i]:Ac.indptr[i + 1]]]) for i in range(A.shape[0])], dtype=np.longdouble)
    xr = xr + lu.solve(rr.astype(np.float64)).astype(np.longdouble)
ref = xr.astype(np.float64)
print("oracle spsolve vs refined:", np.abs(Po.ravel() - ref).max() / np.abs(ref).max())
for v in (1, 2, 4, 5, 8, 0, 7, 9):
    plan = ForwardPlan(gm, 1, 0.025, 1, keep_history=True, device=0)
    plan.set_variant(v, v)
    plan.set_inputs(perm_transf(x), None, transformed=True)
    plan.set_field("S", S[None, :])
    plan.pressure_only(0)
    P = plan.get_field("P")[0].ravel()
    print(f"variant {v}: max|P - refined| / max|P| = {np.abs(P - ref).max() / np.abs(ref).max():.3e}")
    plan.close()
