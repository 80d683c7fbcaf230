"""Batched objective of the production-optimisation tutorial: the GPU replacement of ``apply(obj, U)``.

The reference evaluates an ensemble of control vectors by mapping ``npv(model, **params)`` over the members
(notebooks/Optimise.py:112-125 through ``utils.apply`` at Optimise.py:259, 441, 514, 655): every member re-configures
a copy of the base model (well positions and/or rates, ``remake`` Optimise.py:130-135), simulates ``nTime`` steps and
turns the producers' saturations into a net present value (``accounting`` Optimise.py:170-200); a member whose
configuration is invalid (a well outside the domain, unbalanced rates) is worth 0 (the ``except`` branch,
Optimise.py:119-124).  Here the members run as ONE device batch with a different source field per member
(``hm_fwd_set_member_wells``); the accounting stays NumPy on the (N, nTime+1, nPrd) producer saturations.

Members of a batch may differ in their wells -- positions, rates and their NUMBER; permeability is the base model's (or one field per member).
Rates are taken as given: the upstream simulator's ``actual_rates`` (its rate controller) is not part of the reference
repository (parity unpinned for everything inside ``TPFA_ResSim``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .forward import ForwardPlan
from .ressim import ResSim

# Optimise.py:151-162 -- prices "not grounded in reality", the tutorial's values
ONE_YEAR = 0.1


def default_prices(dt):
    return {"inj": 20, "oil": 100, "turbo": 1, "wat": 6, "diffs": 1, "fixed": 0.8 * dt / ONE_YEAR, "/well": 0.3 * dt / ONE_YEAR}


def discounts(dt, nTime):
    return 0.96 ** (dt / ONE_YEAR * np.arange(nTime))  # Optimise.py:162


def accounting(prd_wsats, inj_rates, prd_rates, dt, price, disc, rate0):
    """Ledger of one member's net present value; formulas of Optimise.py:170-200.

    ``prd_wsats`` (nPrd, nTime): water cut at each producer per time INTERVAL (mean of the interval's end points,
    Optimise.py:205-208); ``inj_rates`` / ``prd_rates`` (nWell, nTime); ``disc`` (nTime,) discount factors.
    Revenue and running costs are discounted field-wide volumes (a producer's stream splits into water and oil by its
    water cut); the remaining terms are penalties on the controls themselves: a charge per (well, interval) in operation,
    a quadratic charge on field production above ``rate0``, and a charge on every change of an injection rate."""
    step_w = dt * np.asarray(disc)                                # discounted length of every interval
    water_cut = np.asarray(prd_wsats)
    produced = np.asarray(prd_rates).sum(axis=0)                  # field totals per interval
    produced_water = (np.asarray(prd_rates) * water_cut).sum(axis=0)
    injected = np.asarray(inj_rates).sum(axis=0)
    in_operation = np.count_nonzero(prd_rates), np.count_nonzero(inj_rates)
    over_capacity = np.maximum(produced - rate0, 0.0).sum()
    rate_changes = np.abs(np.asarray(inj_rates)[:, 1:] - np.asarray(inj_rates)[:, :-1])
    return {
        "oil": price["oil"] * float((produced - produced_water) @ step_w),
        "wat": -price["wat"] * float(produced_water @ step_w),
        "inj": -price["inj"] * float(injected @ step_w),
        "pwell": -price["/well"] * in_operation[0],
        "iwell": -price["/well"] * in_operation[1],
        "turbo": -price["turbo"] * dt * over_capacity**2,
        "diffs": -price["diffs"] * float((rate_changes**0.1).sum()),
    }


def _member_config(model: ResSim, params, nTime):
    """Well cells and (nWell, nTime) rates of ``remake(model, **params)``; raises like the model would when run."""
    inj_xy = params.get("inj_xy", model.inj_xy)
    prd_xy = params.get("prd_xy", model.prd_xy)
    inj_xy = np.asarray(inj_xy, dtype=float).reshape(-1, 2)
    prd_xy = np.asarray(prd_xy, dtype=float).reshape(-1, 2)
    inj = np.asarray(params.get("inj_rates", model.inj_rates), dtype=float).reshape(len(inj_xy), -1)
    prd = np.asarray(params.get("prd_rates", model.prd_rates), dtype=float).reshape(len(prd_xy), -1)
    for r in (inj, prd):
        if r.shape[1] not in (1, nTime):
            raise ValueError("rates must have 1 or nTime columns")
    inj = np.broadcast_to(inj, (len(inj_xy), nTime))
    prd = np.broadcast_to(prd, (len(prd_xy), nTime))
    if not np.allclose(inj.sum(0), prd.sum(0)):
        raise ValueError("total injection rate must equal total production rate")  # HistoryMatch.py:182-184
    inj_ind = model.xy2ind(inj_xy[:, 0], inj_xy[:, 1])  # raises outside the domain (Optimise.py:549-554)
    prd_ind = model.xy2ind(prd_xy[:, 0], prd_xy[:, 1])
    return inj_ind, inj, prd_ind, prd


class NpvBatch:
    """``values = NpvBatch(model, dt, nTime)(list_of_params)`` == ``[npv(model, **params)[0] for params in ...]``.

    ``perms``: None (the base model's permeability for every member, Optimise.py:69), or ``(N, Nxy)`` permeabilities
    (one per member: the robust objectives over an uncertainty ensemble, Optimise.py:908, 1006-1011)."""

    def __init__(self, model: ResSim, dt, nTime, wsat0=None, price=None, rate0=1.5):
        self.model, self.dt, self.nTime = model, float(dt), int(nTime)
        self.wsat0 = np.zeros(model.Nxy) if wsat0 is None else np.asarray(wsat0, dtype=float)
        self.price = default_prices(self.dt) if price is None else dict(price)
        self.disc = discounts(self.dt, self.nTime)
        self.rate0 = float(rate0)
        self.last = None
        self._plan, self._plan_key = None, None  # the device plan is kept between calls of the same batch shape (EnOpt iterations)

    def close(self):
        if self._plan is not None:
            self._plan.close()
            self._plan = None

    __del__ = close

    def as_objective(self, make_params):
        """A per-member objective ``obj(u)`` for ``utils.apply(obj, U)`` (Optimise.py:259, 441, 514, 655): ``make_params(u)`` gives
        the member's model parameters (``dict(inj_xy=...)``, ``dict(inj_rates=...)`` ...).  ``apply`` finds the batched form on
        it and values the whole ensemble of controls in one device run."""
        def obj(u):
            return float(self([make_params(u)])[0])

        obj.batched = lambda U: list(self([make_params(u) for u in U]))
        return obj

    def __call__(self, params_list, perms=None):
        m, nT = self.model, self.nTime
        N = len(params_list)
        if N == 0:
            return np.zeros(0)
        nInj = nPrd = None
        cfgs, valid = [], np.ones(N, dtype=bool)
        for n, params in enumerate(params_list):
            try:
                cfg = _member_config(m, params, nT)
            except Exception:  # invalid model params => penalised member (Optimise.py:119-124)
                valid[n] = False
                cfg = None
            cfgs.append(cfg)
            if cfg is not None:  # members may differ in their number of wells (Optimise.py:736-767 varies the wells per member): the device
                nInj = max(nInj or 0, len(cfg[0]))  # sees a source FIELD per member; the plan is shaped for the largest counts
                nPrd = max(nPrd or 0, len(cfg[2]))
        values = np.zeros(N)
        if not valid.any():
            return values
        steady = all(c is None or ((c[1] == c[1][:, :1]).all() and (c[3] == c[3][:, :1]).all()) for c in cfgs)
        cols = 1 if steady else nT  # one source field per member, or one per member and time step
        q_all = np.zeros((N, cols, m.Nxy))
        prd_all = np.zeros((N, nPrd), dtype=np.int32)
        for n, cfg in enumerate(cfgs):
            if cfg is None:
                continue  # q = 0: the member takes no step and is flagged by the device; its value stays 0
            inj_ind, inj, prd_ind, prd = cfg
            np.add.at(q_all[n], (slice(None), inj_ind), inj.T[:cols])      # q[inj] += rate
            np.subtract.at(q_all[n], (slice(None), prd_ind), prd.T[:cols])  # q[prd] -= rate
            prd_all[n, :len(prd_ind)] = prd_ind
            prd_all[n, len(prd_ind):] = prd_ind[0]  # (a member with fewer producers: its spare gather slots repeat one of its cells; unused)
        key = (N, nInj, nPrd)
        if self._plan is None or self._plan_key != key:
            self.close()
            # a plan with placeholder wells (the per-member ones replace them), same fluid / porosity / dtype
            base = ResSim(m.Nx, m.Ny, m.Lx, m.Ly, dtype=m.dtype, device=m.device)
            for a in ("vw", "vo", "swc", "sor", "por"):
                setattr(base, a, getattr(m, a))
            base.inj_xy, base.prd_xy = [[m.Lx / 2, m.Ly / 2]] * nInj, [[m.Lx / 2, m.Ly / 2]] * nPrd
            base.inj_rates, base.prd_rates = np.ones((nInj, 1)) / nInj, np.ones((nPrd, 1)) / nPrd
            self._plan, self._plan_key = ForwardPlan(base, N, self.dt, nT, keep_history=True), key
        plan = self._plan
        _lib.check(plan.lib.hm_fwd_set_member_wells(plan.h, q_all.ctypes.data_as(C.c_void_p), cols, prd_all.ctypes.data_as(C.c_void_p)),
                   "hm_fwd_set_member_wells")
        K = np.broadcast_to(np.asarray(m.K[0], dtype=float).reshape(1, -1), (N, m.Nxy)) if perms is None else np.asarray(perms, dtype=float)
        plan.set_inputs(np.ascontiguousarray(K), np.broadcast_to(self.wsat0, (N, m.Nxy)), transformed=True)
        plan.run()
        plan.sync()
        wsats, _, status = plan.outputs()
        ok = valid & (status == 0)
        for n in np.flatnonzero(ok):
            inj_ind, inj, prd_ind, prd = cfgs[n]
            s = wsats[n][:, prd_ind]                       # prd_sats, Optimise.py:205-208
            ledger = accounting(((s[:-1] + s[1:]) / 2).T, inj, prd, self.dt, self.price, self.disc, self.rate0)
            values[n] = sum(ledger.values())
        self.last = {"wsats": wsats, "status": status, "valid": valid}
        return values
