"""Cycle stamps of the nested-dissection pressure kernels (block 0, wave 0).  Build first: diag/build_nd_prof.sh, then
     HM_AMD_LIB=build_prof/libhm_ndprof.so python profiles/diag/nd_prof.py [N=1000]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from helpers import make_models, perms  # noqa: E402
from historymatching_amd import _lib  # noqa: E402
from historymatching_amd.forward import ForwardPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
_, gm = make_models(128, 128)
plan = ForwardPlan(gm, N, 0.025, 4, keep_history=False, device=0)
import os  # noqa: E402
if os.environ.get("HM_TOP_DEAL"):
    plan.set_debug("top_deal", int(os.environ["HM_TOP_DEAL"]))
plan.set_variant(14, 0)  # every front eliminated (12 would skip the dry ones: on the initial state nearly all)
plan.set_inputs(perms(128, 128, N, seed=1), None, transformed=False)
for _ in range(4):
    plan.pressure_only(0)
st = plan.sync()
print(f"pressure launch avg {st['ms_pressure'] / st['n_pressure_launches']:.2f} ms for {N} members")
lib = _lib.load()
buf = (C.c_longlong * 64)()
lib.hm_debug_nd_prof.argtypes = [C.POINTER(C.c_longlong)]
assert lib.hm_debug_nd_prof(buf) == 0
top = ["between fronts", "tables + child 0 -> LDS, tile decode, barrier", "coefficients + child 0 gather", "child 1 -> LDS (2 barriers)", "child 1 gather", "S1 sweep", "barrier 1", "S2 W = P V, publish",
       "barrier 2", "S3 updates", "store update", "end barrier", "next pivot tile: update + sweep (its owner)", "next front's tables + children: DMA issue", "(count) panels whose next pivot tile this wave owns"]
v = list(buf[:16])
print(f"k_nd_top, block 0 wave 0: {sum(v)} cycles")
for n, x in zip(top, v):
    print(f"   {n:28s} {x:10d}  {100 * x / max(sum(v), 1):5.1f} %")
lv = list(buf[48:54])
# (round 5: level 4 is a launch of its own -- one front per workgroup of 8 waves, two workgroups a CU; the stamps are those of the kernel that
# ran last, the member-per-workgroup kernel of levels 3..0)
print("   by level (3, 2, 1, 0): " + ", ".join(f"{lv[i - 1] - lv[i]}" for i in (3, 2, 1)) + f", {lv[5] - lv[0]} cycles")
sub = ["staging (leaf updates, recipes, coefficients)", "level 9, first front", "level 9, second front", "level 8 front"]
v = list(buf[16:32])
print(f"k_nd_sub, one workgroup (block 0, or -DHM_ND_PROF_SUB_BLOCK=n: one that starts on a busy GPU), wave 0: {sum(v)} cycles")
for n, x in zip(sub, v):
    print(f"   {n:28s} {x:10d}  {100 * x / max(sum(v), 1):5.1f} %")
v = list(buf[32:48])
print(f"k_nd_solve, block 0 wave 0: {sum(v)} cycles; slots (level 5..10 compute at 5..10, barrier waits at 0..5, levels 0-4 at 12): {v}")
v = list(buf[54:60])
print("k_nd_wave<7>, <6>, <5>, one wave (workgroup HM_ND_PROF_SUB_BLOCK, wave 0): staging (recipes, children by LDS-DMA, coefficient planes) / the front, cycles: "
      + ", ".join(f"{v[2 * i]} / {v[2 * i + 1]}" for i in range(3)))
