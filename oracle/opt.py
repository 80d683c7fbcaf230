"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the optimisation tutorial's objective for the parity tests.

``npv(model, **params)`` as in notebooks/Optimise.py:112-125: copy the model, set the parameters (``remake``
Optimise.py:130-135), simulate with the oracle simulator (oracle/ressim.py), account (Optimise.py:170-200, restated
below from the reference's formulas), return 0 for an invalid configuration.  The ACCOUNTING is pinned: tests/golden/
f10_npv_accounting.npz holds the ledgers the reference's own `accounting` / `prd_sats` (AST-extracted by oracle/make_golden_npv.py)
give on fixed inputs.  The simulator it runs on is unpinned, and rates are taken as given (the upstream
``actual_rates`` controller is not available)."""
import copy

import numpy as np

ONE_YEAR = 0.1


def prices(dt):
    return {"inj": 20, "oil": 100, "turbo": 1, "wat": 6, "diffs": 1, "fixed": 0.8 * dt / ONE_YEAR, "/well": 0.3 * dt / ONE_YEAR}


def accounting(prd_wsats, inj_rates, prd_rates, dt, nTime, rate0=1.5):
    """The ledger of Optimise.py:170-200 (pinned by tests/golden/f10_npv_accounting.npz: the reference's own function on fixed inputs).
    prd_wsats (nPrd, nTime): interval means of the producers' saturations (prd_sats, Optimise.py:205-208); rates (nWell, nTime)."""
    price = prices(dt)
    disc = 0.96 ** (dt / ONE_YEAR * np.arange(nTime))
    inj_total = (dt * inj_rates).sum(0) @ disc
    oil_total = (dt * prd_rates * (1 - prd_wsats)).sum(0) @ disc
    wat_total = (dt * prd_rates * prd_wsats).sum(0) @ disc
    excess = (prd_rates.sum(0) - rate0).clip(0)
    return {"oil": price["oil"] * oil_total, "inj": -price["inj"] * inj_total, "wat": -price["wat"] * wat_total,
            "pwell": -price["/well"] * np.sum(prd_rates != 0), "iwell": -price["/well"] * np.sum(inj_rates != 0),
            "turbo": -price["turbo"] * excess.sum() ** 2 * dt, "diffs": -price["diffs"] * (np.abs(np.diff(inj_rates, 1)) ** 0.1).sum()}


def npv(model, dt, nTime, wsat0, rate0=1.5, **params):
    try:
        model = copy.deepcopy(model)
        for k, v in params.items():
            setattr(model, k, v)
        wsats = model.sim(dt, nTime, wsat0)
        s = wsats[:, model.xy2ind(*model.prd_xy.T)]
        prd_wsats = ((s[:-1] + s[1:]) / 2).T
        inj_rates = np.broadcast_to(np.asarray(model.inj_rates, float).reshape(model.nInj, -1), (model.nInj, nTime))
        prd_rates = np.broadcast_to(np.asarray(model.prd_rates, float).reshape(model.nPrd, -1), (model.nPrd, nTime))
        return sum(accounting(prd_wsats, inj_rates, prd_rates, dt, nTime, rate0).values()), wsats
    except Exception:
        return 0, None
