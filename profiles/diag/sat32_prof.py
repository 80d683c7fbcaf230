"""Cycle stamps of the float32 register sweep on slabs (k_sat32s, one workgroup, every wave), one launch per time step of a forward run.
Build first: diag/build_sat32_prof.sh [workgroup id], then
     HM_AMD_LIB=build_prof/libhm_sat32prof.so python profiles/diag/sat32_prof.py [grid=128] [steps=0,5,20,39] [N=256]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,5,20,39").split(",")]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
_, gm = make_models(grid, grid, dtype=32)
plan = ForwardPlan(gm, N, 0.025, 40, keep_history=False, device=0)
plan.set_inputs(perms(grid, grid, N, seed=1), None, transformed=False)
lib = _lib.load()
lib.hm_debug_sat32_prof.argtypes = [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 64)()
names = ["publish", "barrier", "halo/polls", "sweep", "fold"]
for k in range(max(steps) + 1):
    plan.run(k, 1)
    if k not in steps:
        continue
    st = plan.sync()
    assert lib.hm_debug_sat32_prof(buf) == 0
    v = [list(buf[8 * w:8 * w + 8]) for w in range(8)]
    nts = v[0][7]
    print(f"time step {k}: Nts {nts}, loop {v[0][5] / nts:.0f} cycles per sub-step; saturation launch {st['ms_saturation'] / st['n_saturation_launches']:.2f} ms avg so far")
    for w in range(8):
        print(f"   wave {w}: dry {v[w][6]:5d} of {nts}   " + "   ".join(f"{n} {v[w][i] / nts:7.0f}" for i, n in enumerate(names)))
plan.close()
