"""Per-kernel, per-grid launch averages of the larger grids' nested dissection from a rocprofv3 kernel trace (csv) of tests/tools/ndl_check.py:
    python3 profiles/tools/ndl_levels.py <dir> [members]"""
import csv
import glob
import re
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(s in n for s in ("k_nd", "k_big", "k_ndl")):
        continue
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    if wg and grid // wg < N:
        continue  # the three-member accuracy runs
    short = re.search(r"k_\w+(<[^>]*>)?", n).group(0)
    acc[(short, grid // max(wg, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
rows = []
for (n, wgs), v in acc.items():
    v = sorted(v)
    med = v[len(v) // 2]
    rows.append((n, wgs, len(v), med))
per_step = defaultdict(float)
calls = max(c for _, _, c, _ in rows)
for n, wgs, c, med in sorted(rows, key=lambda t: -t[3] * t[2]):
    print(f"{n:28s} workgroups {wgs:9d} calls {c:4d} median {med:9.1f} us")
    per_step[n.split('<')[0]] += med * c
print("per time step (ms):", {k: round(v / 6 / 1e3, 2) for k, v in sorted(per_step.items(), key=lambda t: -t[1])}, "sum", round(sum(per_step.values()) / 6 / 1e3, 2))
