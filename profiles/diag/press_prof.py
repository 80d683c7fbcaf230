"""Cycle breakdown of k_press128m (workgroup 0, thread 0).  Build the instrumented library first:
     make -C historymatching_amd/csrc clean && make -C historymatching_amd/csrc EXTRA=-DHM_PRESS_PROF TARGET=/tmp/libhm_prof.so
   then  HM_AMD_LIB=/tmp/libhm_prof.so python profiles/diag/press_prof.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402
from historymatching_amd.geostat import gaussian_fields_kron  # noqa: E402

n_e = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = bench.build_model(64, device=0)
perms = gaussian_fields_kron(128, 128, 2, 1, n_e, r=0.8, seed=1)
plan = ForwardPlan(model, n_e, bench.DT, 4, keep_history=True, device=0)
plan.set_variant(variant, variant)
plan.set_inputs(perms, None, transformed=False)
plan.run(0, 4)
st = plan.sync()
lib = _lib.load()
if variant not in (3, 4, 5, 8):  # press128s: compute wave 1 and the sweeper
    buf = (C.c_longlong * 32)()
    lib.hm_debug_press_prof_s.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.hm_debug_press_prof_s(buf) == 0
    names = ["B: W tiles", "barrier Y", "C pass 1 (next column, publish)", "C pass 2", "barrier X", "assembly", "vectors -> LDS + barrier",
             "mat-vec + scale", "D add + publish 0", "barrier (sweep 0)", "G store + barrier", "back substitution", "face fluxes",
             "sweeper: wait for tile", "sweeper: sweep", "-"]
    print(f"pressure launch avg {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms for {n_e} members")
    for who, off in (("compute wave 1", 0), ("sweeper (wave 0)", 16)):
        v = np.array(buf[off:off + 16], dtype=np.float64)
        print(f" {who}: total {v.sum():.0f} cycles")
        for n, x in zip(names, v):
            if x:
                print(f"   {n:34s} {x:12.0f} cycles  {100 * x / v.sum():5.1f} %   per panel {x / 1024:7.0f}")
    sys.exit(0)
buf = (C.c_longlong * 16)()
lib.hm_debug_press_prof.argtypes = [C.POINTER(C.c_longlong)]
assert lib.hm_debug_press_prof(buf) == 0
names = ["A: publish+inverse", "barrier 1", "W = U P", "barrier 2", "rank-16 update", "assembly", "per-ix preamble",
         "G store + tail", "back substitution", "face fluxes",
         "  pre: vec->LDS + barrier", "  pre: matvec", "  pre: scale", "-", "-", "-"]
v = np.array(buf[:16], dtype=np.float64)
print(f"pressure launch avg {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms for {n_e} members")
for n, x in zip(names, v):
    print(f"  {n:22s} {x:12.0f} cycles  {100 * x / v.sum():5.1f} %")
print(f"  total {v.sum():.0f} cycles; per panel (1024 panels): {v[:5].sum() / 1024:.0f}")
