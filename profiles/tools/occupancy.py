#!/usr/bin/env python3
"""Achieved occupancy per kernel from one rocprofv3 counter pass (SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE):
    python3 profiles/tools/occupancy.py <counter_collection.csv> [name filter]
waves per CU ~ SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES (wave-cycles spent resident over cycles a CU had at least one wave)."""
import collections
import csv
import re
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if flt not in n:
        continue
    mm = re.search(r"(k_\w+(<[^>]*>)?)", n)
    k = mm.group(1) if mm else n[:28]
    acc[k[:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    g = lambda c: m.get(c, 0.0)  # noqa: E731
    print(f"{k:28s} waves {g('SQ_WAVES'):9.0f}  wave-cycles {g('SQ_WAVE_CYCLES'):.3e}  busy CU-cycles {g('SQ_BUSY_CU_CYCLES'):.3e}  GUI active {g('GRBM_GUI_ACTIVE'):.3e}"
          f"  -> {g('SQ_WAVE_CYCLES') / max(g('SQ_BUSY_CU_CYCLES'), 1.0):5.2f} waves per busy CU, {g('SQ_WAVE_CYCLES') / max(g('SQ_WAVES'), 1.0):9.0f} cycles per wave")
