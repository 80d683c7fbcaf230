"""HBM bytes per member and time step of the larger grids' nested-dissection pressure step from two rocprofv3 PMC passes (FETCH_SIZE,
WRITE_SIZE; KiB on gfx950; same caveat as pmc_to_json.py: FETCH_SIZE counts half of the bytes of wide coalesced reads -- raw and x2 given):
   python3 profiles/tools/pmc_nd_large_to_json.py <dir with pmc_nd256_{FETCH,WRITE}_SIZE.csv> <members>"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

d, members = Path(sys.argv[1]), int(sys.argv[2])


def totals(counter):
    acc, calls = defaultdict(float), defaultdict(int)
    with open(d / f"pmc_nd256_{counter}.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = re.search(r"k_\w+", row["Kernel_Name"]).group(0)
            acc[k] += float(row["Counter_Value"]) * 1024.0
            calls[k] += 1
    return acc, calls


fetch, calls = totals("FETCH_SIZE")
write, _ = totals("WRITE_SIZE")
steps = max(1, calls.get("k_nd_flux", 1))  # one k_nd_flux launch per time step
out, tot_raw, tot_x2 = {}, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) / steps / members, write.get(k, 0.0) / steps / members
    out[k] = {"launches_per_step": calls.get(k, 0) / steps, "fetch_bytes_raw_per_member_step": f, "write_bytes_per_member_step": w}
    if k != "k_sat256s":
        tot_raw += f + w
        tot_x2 += 2 * f + w
print(json.dumps({"members": members, "time_steps": steps, "kernels": out,
                  "pressure_step_hbm_bytes_per_member_raw": tot_raw, "pressure_step_hbm_bytes_per_member_fetch_x2": tot_x2}, indent=1))
