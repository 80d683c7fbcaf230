"""Per-kernel HBM bytes from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; both in KiB on gfx950).

gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts half of the bytes of wide
coalesced reads; calibrated on k_perm_transform (reads 128 KiB per member: N * 131072 B expected).  The x2 is applied
to the kernels whose reads are 16 B/lane streams (perm_transform, press128m); sat128's fetches are dominated by dword
scratch reloads and are reported uncorrected, with the x2 figure as a bracket."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out_dir, members = Path(sys.argv[1]), int(sys.argv[2])
KEYS = {"k_perm_transform": "perm_transform", "k_press128m": "press128m", "k_press128s": "press128s", "k_pressure_pcg": "pressure_pcg", "k_press128<": "press128", "k_sat128": "sat128",
        "k_pressure_generic": "pressure_generic", "k_saturation_generic": "saturation_generic",
        "k_nd_assemble": "nd_assemble", "k_nd_sub(": "nd_sub", "k_nd_wave<7": "nd_wave7", "k_nd_wave<6": "nd_wave6", "k_nd_wave<5": "nd_wave5",
        "k_nd_top<3, 5": "nd_top4", "k_nd_top<3, 4": "nd_top",  # (level 4: a launch of its own, one front per workgroup; levels 3..0 -- rounds 3-5 averaged the two under one key)
         "k_nd_solve(": "nd_solve", "k_nd_solve_sub": "nd_solve_sub", "k_nd_flux": "nd_flux", "k_nd_leaf(": "nd_leaf", "k_nd_leaf_solve": "nd_leaf_solve"}


def per_launch(counter):
    acc = defaultdict(list)
    with open(out_dir / f"pmc_{counter}_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for pat, key in KEYS.items():
                if pat in row["Kernel_Name"]:
                    acc[key].append(float(row["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    # round 6: the nested-dissection kernels read contiguous 512-byte wave rows (8 bytes a lane).  The guide calls that width uncalibrated;
    # calibrated here on k_nd_solve, whose compulsory stream is known -- the factor of levels 0..7, 304 896 doubles = 2.44 MB per member,
    # every byte read once -- against 1.43 MB raw: the counter tallies these reads at one half as well (x 2 = 2.87 MB = the factor + the
    # gathers of the boundary pressures).  So the x 2 applies to every nd_* kernel's fetches.
    wide = k in ("perm_transform", "press128m", "press128s") or k.startswith("nd_")
    kernels[k] = {
        "fetch_bytes_raw_per_launch": f, "write_bytes_per_launch": w,
        "hbm_bytes_per_member_corrected": ((2.0 * f if wide else f) + w) / members,
        "hbm_bytes_per_member_uncorrected": (f + w) / members,
        "hbm_bytes_per_member_fetch_x2": (2.0 * f + w) / members,
    }
nd_keys = [k for k in kernels if k.startswith("nd_")]
if nd_keys:  # the nested-dissection pressure solve as a whole (its launches of one time step together)
    kernels["press_nd"] = {kk: sum(kernels[k][kk] for k in nd_keys) for kk in kernels[nd_keys[0]]}
    # (the sum of the corrected per-kernel figures: rounds 3-5 reported the uncorrected sum here)
    kernels["press_nd"]["launches"] = nd_keys
print(json.dumps({
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --members %d --steps 1; gfx950 "
            "correction: FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads (x2 applied to press128s/press128m and "
            "perm_transform, 16 B/lane streams; calibration: k_perm_transform reads 131072 B per member; from round 6 on also to the nd_* kernels, "
            "8 B/lane rows of 512 B, calibrated on k_nd_solve's factor stream: 2.44 MB known against 1.43 MB raw); sat128 "
            "uncorrected (dword scratch reloads), x2 figure given as a bracket" % members,
    "members_per_launch": members, "kernels": kernels}, indent=1))
