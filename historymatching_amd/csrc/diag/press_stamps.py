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
names = ["frags", "la: issue LA MFMA", "la: wait+fix LA tiles", "la: publish(+inv4)", "la: rest MFMA+fix", "other: issue MFMA",
         "other: fix (+MFMA wait)", "la: barrier wait", "other: barrier wait", "n la panels", "n other panels", "kernel total"]
for w, off in ((0, 0), (5, 16), (3, 32)):
    v = dbg[off:off + 16]
    nla, no = max(int(v[9]), 1), max(int(v[10]), 1)
    print("wave", w, "la panels", nla, "other panels", no, "kernel ticks", int(v[11]))
    print("   frags/panel", int(v[0] / (nla + no)))
    print("   la panel:   ", {n: int(v[i] / nla) for i, n in ((1, names[1]), (2, names[2]), (3, names[3]), (4, names[4]), (7, names[7]))})
    print("   other panel:", {n: int(v[i] / no) for i, n in ((5, names[5]), (6, names[6]), (8, names[8]))})
print("ms for 2 launches", st["ms_pressure"])
