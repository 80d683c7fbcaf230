"""TEST INFRASTRUCTURE ONLY -- fixture F9: the arrays the reference's OWN script produces when it runs end to end.

Executes notebooks/HistoryMatch.py (BASELINE config 1: "Reference HistoryMatch.py CPU run") exactly as oracle/run_reference_script.py does
-- the real script text from /root/reference, with oracle/ressim.py standing in for the absent `TPFA_ResSim` package and no-op
stand-ins for the plotting-only packages -- and stores what it computed in tests/golden/f9_hm_script.npz: data only, the script never
travels.  Two harness-level edits of the executed text, nothing else: `utils.nCPU = 1` (no process pool) and one inserted line that keeps
the IES call's `stats` before the ILES call overwrites the name.

Contents (HistoryMatch.py line of the assignment):
  perm_Truth (:167), perm_Prior (:290), prod_past_Truth / prod_past_Noisy (:224-267), wsat_past_Truth, prod_past_Prior, wsat_final_Prior (:400-401),
  obs, perturbs, decorr, obs_ens = hm_setup0 (:635-640), perm_ES (:652), taper_LES = bump(distances_to_obs / 1.2), perm_LES (:863),
  perm_IES + the simulated observations of its 10 iterates (:958-961), perm_ILES + its iterates' observations (:1075-1077),
  model: Nx, Ny, Lx, Ly, dt, nTime, wells.
The simulator half of these numbers is the ORACLE's (parity unpinned); the update half -- every perm_* given the stored
observations -- is the reference's own arithmetic on them.

Usage:  python oracle/make_golden_script.py        (~3 min on one core; /root/reference must exist)"""
import re
import sys
import time
import types
from pathlib import Path
from unittest import mock

import matplotlib
import numpy as np

matplotlib.use("Agg")
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = "/root/reference/notebooks"
if not Path(REF).exists():
    raise SystemExit("/root/reference not present: this fixture can only be regenerated in the build container")
sys.path.insert(0, REF)


class Anything(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return mock.MagicMock(name=f"{self.__name__}.{k}")


for name in ("adjustText", "mpl_tools", "mpl_tools.place", "mpl_tools.misc", "ipywidgets", "IPython", "IPython.display"):
    sys.modules[name] = Anything(name)
sys.modules["mpl_tools.misc"].nRowCol = lambda *a, **k: {"nrows": 1, "ncols": 1}
st = types.ModuleType("struct_tools")


class DotDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def __dir__(self):
        return list(self.keys())


st.DotDict = DotDict
sys.modules["struct_tools"] = st

import oracle.ressim as oressim  # noqa: E402


class ResSim(oressim.ResSim):
    def __getattr__(self, k):
        if k.startswith("plt_") or k == "anim":
            return lambda *a, **kw: mock.MagicMock()
        raise AttributeError(k)


sim = types.ModuleType("TPFA_ResSim")
sim.ResSim = ResSim
simp = types.ModuleType("TPFA_ResSim.plotting")
simp.styles = mock.MagicMock()
sys.modules["TPFA_ResSim"] = sim
sys.modules["TPFA_ResSim.plotting"] = simp


class PlotStub(Anything):
    @staticmethod
    def freshfig(*a, ncols=1, nrows=1, **k):
        n = ncols * nrows
        return mock.MagicMock(), (mock.MagicMock() if n == 1 else tuple(mock.MagicMock() for _ in range(n)))

    @staticmethod
    def figure12(*a, **k):
        return mock.MagicMock(), [mock.MagicMock() for _ in range(3)]


import tools  # noqa: E402

plotting = PlotStub("tools.plotting")
plotting.styles = {"oil": {}, "pperm": {"levels": np.linspace(-4, 4, 21)}, "corr": {}, "NPV": {}}
sys.modules["tools.plotting"] = plotting
tools.plotting = plotting
del sys.modules["IPython"], sys.modules["IPython.display"]
import tools.utils as utils  # noqa: E402

utils.nCPU = 1
src = open(f"{REF}/HistoryMatch.py").read()
src = re.sub(r"utils\.nCPU = .*", "utils.nCPU = 1", src)
src, n_ins = re.subn(r"(perm\.IES, stats = IES\(.*\)\n)", r"\1stats_IES = stats\n", src)
assert n_ins == 1
t0 = time.time()
ns = {"__name__": "__main__"}
exec(compile(src, "HistoryMatch.py", "exec"), ns)
print("script completed in %.0f s" % (time.time() - t0))

perm, prod, wsat, model = ns["perm"], ns["prod"], ns["wsat"], ns["model"]
hm0 = ns["hm_setup0"]
out = dict(
    Nx=model.Nx, Ny=model.Ny, Lx=model.Lx, Ly=model.Ly, dt=ns["dt"], nTime=ns["nTime"], seed=ns["seed"],
    inj_xy=np.asarray(model.inj_xy, float), prd_xy=np.asarray(model.prd_xy, float),
    inj_rates=np.asarray(model.inj_rates, float), prd_rates=np.asarray(model.prd_rates, float),
    prod_inds=np.asarray(ns["prod_inds"]),
    perm_Truth=perm.Truth, perm_Prior=perm.Prior, perm_ES=perm.ES, perm_LES=perm.LES, perm_IES=perm.IES, perm_ILES=perm.ILES,
    wsat_past_Truth=np.asarray(wsat.past.Truth), prod_past_Truth=np.asarray(prod.past.Truth), prod_past_Noisy=np.asarray(prod.past.Noisy),
    prod_past_Prior=np.asarray(prod.past.Prior), wsat_final_Prior=np.asarray(wsat.past.Prior)[:, -1, :],
    obs=np.asarray(hm0["obs"]), perturbs=np.asarray(hm0["perturbs"]), decorr=np.asarray(hm0["decorr"]), obs_ens=np.asarray(hm0["obs_ens"]),
    taper_LES=ns["loc"].bump(ns["distances_to_obs"] / 1.2),
    IES_Eo=np.asarray(ns["stats_IES"].Eo), ILES_Eo=np.asarray(ns["stats"].Eo),
    IES_xStep=0.4, IES_iMax=10,
)
for k, v in out.items():
    v = np.asarray(v)
    print(f"  {k:18s} {str(v.shape):16s} {v.dtype}")
np.savez_compressed(ROOT / "tests" / "golden" / "f9_hm_script.npz", **out)
print("wrote tests/golden/f9_hm_script.npz", (ROOT / "tests" / "golden" / "f9_hm_script.npz").stat().st_size, "bytes")
