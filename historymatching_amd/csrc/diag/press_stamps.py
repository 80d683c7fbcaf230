"""Diagnostic only (never part of a timed run): per-phase cycle shares of one k_press128m launch.
Build: hipcc ... -DPRESS_STAMPS press128m.hip -> diag/libhm_stamps.so ; run with HM_AMD_LIB pointing at it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
os.environ["HM_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhm_stamps.so")
import numpy as np
from historymatching_amd.forward import ForwardPlan
from tests.helpers import make_models, perms

_, gm = make_models(128, 128)
plan = ForwardPlan(gm, 1, 0.025, 1)
plan.set_inputs(perms(128, 128, 1), transformed=False)
plan.pressure_only(0)
plan.pressure_only(0)
st = plan.sync()
tx = plan.get_field("TX").ravel()
dbg = tx.view(np.int64)[-64:]
names = ["A publish", "barrier1", "B pivot-block inverse", "barrier2", "C frags+MFMA issue", "D fix-ups (role panels)", "D (no role)", "kernel total"]
for w, off in ((0, 0), (5, 8), (3, 16)):
    v = dbg[off:off + 8]
    print("wave", w, {n: int(x) for n, x in zip(names, v)})
    tot = v[:7].sum()
    print("   per panel:", {n: int(x / (128 * 32)) for n, x in zip(names[:7], v[:7])}, "sum", int(tot / (128 * 32)), " kernel cycles", int(v[7]))
print("ms for 2 launches", st["ms_pressure"])
