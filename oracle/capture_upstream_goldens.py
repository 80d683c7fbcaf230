#!/usr/bin/env python3
"""Pin the simulator oracle the day the upstream package is importable ("the first action").

The simulator arithmetic of the reference lives in the un-vendored dependency ``TPFA-ResSim @ adc89536``
(/root/reference/requirements.txt:1; imported as ``simulator`` at notebooks/HistoryMatch.py:88).  It is absent from this
image, so ``oracle/ressim.py`` -- a restatement of the published algorithm the reference cites (HistoryMatch.py:93-95) -- is
UNPINNED: nothing the reference ships fixes a single simulator number.  This script closes that gap wherever
``import TPFA_ResSim`` works (``pip install git+https://github.com/patnr/TPFA-ResSim.git@adc89536``):

  1. builds the reference's truth case exactly as notebooks/HistoryMatch.py:97-224 does (20 x 20 grid on 2 x 1, four producers
     near the corners at rate 1/4, one central injector at rate 1, dt = 0.025, nTime = 40, wsat0 = 0) on the truth
     permeability of the seed-1 RNG replay already committed as fixture F1 (tests/golden/f1_rng_replay.npz: ``perm_truth``),
  2. runs the UPSTREAM ``model.sim`` and stores inputs + outputs as tests/golden/f8_upstream_sim.npz (data only),
  3. runs ``oracle/ressim.py`` on the same inputs and prints / stores the differences: saturation history, producer series,
     per-step sub-step counts where upstream exposes them, the grid helpers (xy2ind / ind2xy / mesh) and ``actual_rates``,
  4. exits 0 if the restatement reproduces upstream to 1e-9 (two sparse direct solves of the same systems), 1 otherwise --
     in which case the listed differences say which of SURVEY.md Appendix A's [U] assumptions (CFL constant, fluid defaults,
     index clamping, rate handling) is wrong.

Without the package it says so and exits 2; nothing is written.  Test infrastructure: never imported by the product."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def build_truth_case(cls):
    """notebooks/HistoryMatch.py:97, 177-190."""
    model = cls(Nx=20, Ny=20, Lx=2, Ly=1)
    near01 = np.array([0.12, 0.87])
    model.prd_xy = [[x, y] for y in model.Ly * near01 for x in model.Lx * near01]
    model.inj_xy = [[model.Lx / 2, model.Ly / 2]]
    model.inj_rates = [[1]]
    model.prd_rates = np.ones((4, 1)) / 4
    return model


def set_perm(model, log_perm):
    """notebooks/HistoryMatch.py:137-138, 160-164."""
    p = (0.1 + np.exp(5 * np.asarray(log_perm))).reshape(model.shape)
    model.K = np.stack([p, p])


def main():
    try:
        import TPFA_ResSim as simulator
    except ImportError as e:
        print(f"TPFA_ResSim is not importable here ({e}); the simulator oracle stays unpinned (DESIGN.md section 2). Nothing written.")
        return 2
    from oracle import ressim as orc

    perm_truth = np.load(ROOT / "tests" / "golden" / "f1_rng_replay.npz")["perm_truth"][0]
    dt, nTime = 0.025, 40
    up = build_truth_case(simulator.ResSim)
    set_perm(up, perm_truth)
    wsat0 = np.zeros(up.Nxy)
    w_up = np.asarray(up.sim(dt, nTime, wsat0, pbar=False))
    prod_inds = np.asarray(up.xy2ind(*np.asarray(up.prd_xy).T))
    fixture = dict(perm_truth=perm_truth, dt=dt, nTime=nTime, wsats=w_up, prods=w_up[1:][:, prod_inds], prod_inds=prod_inds,
                   mesh_x=np.asarray(up.mesh[0]), mesh_y=np.asarray(up.mesh[1]), ind2xy=np.asarray(up.ind2xy(np.arange(up.Nxy))))
    rates = getattr(up, "actual_rates", None)
    if rates is not None:
        fixture.update(actual_inj=np.asarray(rates["inj"]), actual_prd=np.asarray(rates["prd"]))
    out = ROOT / "tests" / "golden" / "f8_upstream_sim.npz"
    np.savez_compressed(out, **fixture)
    print(f"wrote {out.relative_to(ROOT)} (upstream TPFA_ResSim {getattr(simulator, '__version__', '?')})")

    om = build_truth_case(orc.ResSim)
    set_perm(om, perm_truth)
    w_or = om.sim(dt, nTime, wsat0)
    report = {
        "max_abs_diff_saturation": float(np.abs(w_or - w_up).max()),
        "first_step_max_abs_diff": float(np.abs(w_or[1] - w_up[1]).max()),
        "max_abs_diff_producer_series": float(np.abs(w_or[1:][:, prod_inds] - fixture["prods"]).max()),
        "prod_inds_equal": bool(np.array_equal(prod_inds, om.xy2ind(*np.asarray(om.prd_xy).T))),
        "mesh_equal": bool(np.allclose(fixture["mesh_x"], om.mesh[0]) and np.allclose(fixture["mesh_y"], om.mesh[1])),
        "ind2xy_equal": bool(np.allclose(fixture["ind2xy"], om.ind2xy(np.arange(om.Nxy)))),
        "oracle_nts_trace": om.nts_trace.tolist(),
    }
    if rates is not None:
        report["actual_rates_equal"] = bool(np.allclose(fixture["actual_inj"], om.actual_rates["inj"]) and np.allclose(fixture["actual_prd"], om.actual_rates["prd"]))
    (ROOT / "tests" / "golden" / "f8_upstream_vs_oracle.json").write_text(json.dumps(report, indent=1) + "\n")
    print(json.dumps(report, indent=1))
    ok = report["max_abs_diff_saturation"] <= 1e-9 and report["prod_inds_equal"] and report["mesh_equal"]
    print("oracle/ressim.py reproduces upstream: parity PINNED by tests/golden/f8_upstream_sim.npz" if ok else
          "oracle/ressim.py differs from upstream: see SURVEY.md Appendix A, the assumptions marked [U]")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
