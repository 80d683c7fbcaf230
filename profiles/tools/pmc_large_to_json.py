"""Per-kernel HBM bytes of the large-grid path from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB on gfx950) of
tests/tools/large_grid_timing.py.  Same gfx950 caveat as pmc_to_json.py: FETCH_SIZE counts half of the bytes of wide coalesced
reads -- both the raw and the x2 figure are given, with the algorithmic bytes of each kernel beside them.
   python3 profiles/tools/pmc_large_to_json.py <dir with pmc_large_{FETCH,WRITE}_SIZE.csv> <members> <Nx> <Ny>"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

d, members, Nx, Ny = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
KEYS = ["k_coarse_solve", "k_tg_spmv", "k_tg_update", "k_tg_restrict", "k_tg_correct", "k_tg_postsmooth", "k_tg_direction", "k_sat128t", "k_sat256s", "k_press128s",
        "k_tl_setup", "k_tl_final"]


def per_launch(counter):
    acc = defaultdict(list)
    with open(d / f"pmc_large_{counter}.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for key in KEYS:
                if key in row["Kernel_Name"]:
                    acc[key].append(float(row["Counter_Value"]) * 1024.0)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
nxy, c = Nx * Ny, Ny // 128
nxc = Nx // c
vec = 8.0 * nxy  # one fp64 cell vector of a member
alg = {  # algorithmic bytes per member and launch (vector passes of 8 B per cell; the coarse factor: 36 of 64 tiles, two passes)
    "k_coarse_solve": 2 * nxc * 36 * 2048 + 3 * 8.0 * nxc * 128,
    "k_tg_spmv": 4 * vec, "k_tg_update": 7 * vec, "k_tg_restrict": 4 * vec, "k_tg_correct": 2 * vec, "k_tg_postsmooth": 6 * vec,
    "k_tg_direction": 3 * vec, "k_sat128t": 4 * vec, "k_sat256s": 4 * vec,  # compulsory: S, Vx, Vy in, S out (the edge granules come on top: see DESIGN.md)
}
out = {}
for k in KEYS:
    if k in fetch or k in write:
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        per = members
        if k == "k_sat128t":  # a launch holds one round of teams: at most 8 * (32 / tiles) members
            per = min(members, 8 * (32 // ((Nx // 128) * (Ny // 128))))
        if k == "k_sat256s":  # teams of Nx / 64 slabs
            per = min(members, 8 * (32 // (Nx // 64)))
        f, w = f * members / per, w * members / per
        out[k] = {"launches": nf, "members_per_launch": per, "fetch_bytes_raw_per_member": f / members, "write_bytes_per_member": w / members,
                  "hbm_bytes_per_member_raw": (f + w) / members, "hbm_bytes_per_member_fetch_x2": (2 * f + w) / members,
                  "algorithmic_bytes_per_member": alg.get(k)}
print(json.dumps({"note": "averages over all launches of a kernel (members that have converged drop out of the CG kernels, so late "
                          "launches move fewer bytes than the algorithmic figure for one active member)",
                  "members": members, "grid": [Nx, Ny], "kernels": out}, indent=1))
