"""Matrix-core utilisation of the update kernels from one rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_F32/F64,
SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE): per kernel, averaged over its dispatches.
One MOP = 512 flops (rocprof's MFMA FLOP metric); dispatch duration from the counter file's timestamps (counter
collection serialises kernels, so durations are a little longer than in the timing runs)."""
import csv
import json
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
disp = defaultdict(dict)
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"])
    disp[key][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[key]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = defaultdict(list)
for (_, name), c in disp.items():
    mm = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    short = mm.group(1) if mm else name[:60]
    agg[short].append(c)
out = {}
for name, cs in sorted(agg.items()):
    n = len(cs)
    avg = lambda k: sum(c.get(k, 0.0) for c in cs) / n  # noqa: E731
    ns = avg("_ns")
    f32, f64 = avg("SQ_INSTS_VALU_MFMA_MOPS_F32") * 512, avg("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512
    peak = 157.3e12 if f32 >= f64 else 78.6e12
    out[name] = {"dispatches": n, "avg_us_under_pmc": ns / 1e3, "mfma_flops_f32": f32, "mfma_flops_f64": f64,
                 "mfma_tflops": (f32 + f64) / ns / 1e3, "frac_of_matrix_peak": (f32 + f64) / (ns * 1e-9) / peak,
                 "SQ_VALU_MFMA_BUSY_CYCLES": avg("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CU_CYCLES": avg("SQ_BUSY_CU_CYCLES"),
                 "GRBM_GUI_ACTIVE": avg("GRBM_GUI_ACTIVE")}
print(json.dumps(out, indent=1))
