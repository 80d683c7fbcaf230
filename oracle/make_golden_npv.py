"""TEST INFRASTRUCTURE ONLY -- fixture F10: the reference's own NPV accounting on fixed inputs.

`accounting` and `prd_sats` (notebooks/Optimise.py:170-208) and the module-level constants they read (`OneYear`, `price`, `discounts`,
`rate0`: Optimise.py:81, 151-162) are AST-extracted from the real script and executed here on a stand-in model object -- the functions
only touch `model.actual_rates`, `model.prd_xy`, `model.xy2ind` -- with synthetic saturation histories and rate schedules that exercise
every term of the ledger (wells shut in for part of the time, field production above `rate0`, changing injection rates).  Stores
inputs and ledgers in tests/golden/f10_npv_accounting.npz; the script text never travels.  oracle/opt.py and the product's
historymatching_amd/opt.py:accounting are checked against it (tests/test_oracle_golden.py).

Usage:  python oracle/make_golden_npv.py   (/root/reference must exist)"""
import ast
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/notebooks/Optimise.py")
if not REF.exists():
    raise SystemExit("/root/reference not present: this fixture can only be regenerated in the build container")

dt, nTime = 0.025, 40  # HistoryMatch.py:219-221 / Optimise.py: T = 1, dt = 0.025
ns = dict(np=np, dt=dt, nTime=nTime)
tree = ast.parse(REF.read_text())
want_assign = {"OneYear", "price", "discounts", "rate0"}
want_def = {"accounting", "prd_sats"}
seen = set()
for node in tree.body:
    if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id in want_assign \
            and node.targets[0].id not in seen:
        seen.add(node.targets[0].id)
        exec(compile(ast.Module([node], []), "Optimise.py", "exec"), ns)
    if isinstance(node, ast.FunctionDef) and node.name in want_def:
        seen.add(node.name)
        exec(compile(ast.Module([node], []), "Optimise.py", "exec"), ns)
assert seen == want_assign | want_def, seen


class Model:  # what accounting() / prd_sats() touch of the simulator object
    def __init__(self, Nx, Ny, prd_cells, inj, prd):
        self.Nx, self.Ny = Nx, Ny
        self.prd_xy = np.array([[c // Ny + 0.5, c % Ny + 0.5] for c in prd_cells])
        self.actual_rates = {"inj": inj, "prd": prd}

    def xy2ind(self, x, y):
        return (np.floor(x).astype(int) * self.Ny + np.floor(y).astype(int))


rng = np.random.RandomState(10)
Nx = Ny = 12
cases = []
for case in range(6):
    nInj, nPrd = [(1, 4), (2, 3), (1, 1), (3, 2), (1, 4), (2, 2)][case]
    prd_cells = rng.choice(Nx * Ny, nPrd, replace=False)
    wsats = np.sort(rng.rand(nTime + 1, Nx * Ny), axis=0) * rng.rand()  # monotone in time, like a water front
    if case in (0, 2):
        inj = np.full((nInj, nTime), 1.5 / nInj)
        prd = np.full((nPrd, nTime), 1.5 / nPrd)
    else:
        inj = rng.rand(nInj, nTime) * 2.0
        prd = rng.rand(nPrd, nTime) * (1.2 if case != 4 else 0.3)
        inj[:, rng.rand(nTime) < 0.2] = 0.0            # shut-in intervals
        prd[rng.randint(nPrd), : nTime // 3] = 0.0      # a producer that starts late
    m = Model(Nx, Ny, prd_cells, inj, prd)
    ledger = ns["accounting"](m, wsats)
    cases.append(dict(wsats=wsats, prd_cells=prd_cells, inj=inj, prd=prd, ledger=ledger, prd_wsats=ns["prd_sats"](m, wsats).T))
    print(case, {k: round(float(v), 6) for k, v in ledger.items()}, "NPV", round(float(sum(ledger.values())), 6))

keys = list(cases[0]["ledger"].keys())
out = dict(dt=dt, nTime=nTime, rate0=ns["rate0"], ledger_keys=np.array(keys), n_cases=len(cases), Nx=Nx, Ny=Ny,
           price_keys=np.array(list(ns["price"].keys())), price_values=np.array([float(v) for v in ns["price"].values()]),
           discounts=np.asarray(ns["discounts"]))
for i, c in enumerate(cases):
    out[f"wsats_{i}"] = c["wsats"]
    out[f"prd_cells_{i}"] = c["prd_cells"]
    out[f"inj_{i}"] = c["inj"]
    out[f"prd_{i}"] = c["prd"]
    out[f"prd_wsats_{i}"] = c["prd_wsats"]
    out[f"ledger_{i}"] = np.array([float(c["ledger"][k]) for k in keys])
np.savez_compressed(ROOT / "tests" / "golden" / "f10_npv_accounting.npz", **out)
print("wrote tests/golden/f10_npv_accounting.npz")
