#!/usr/bin/env python3
"""fp64 roofline of the forward kernels from one rocprofv3 --pmc pass, reproducible from the committed CSVs:

    python3 profiles/tools/fp64_roofline.py profiles/r02/pmc_fp64_forward_counter_collection.csv \
            profiles/r02/kernel_stats_bench.csv profiles/r02/isa_counts.json profiles/r02/bench_under_rocprof.json \
            > profiles/r02/fp64_roofline.json

Counters (own pass, counters only): SQ_INSTS_VALU, SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
GRBM_GUI_ACTIVE.  Per kernel, averaged over its dispatches:
  * effective clock  = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS give-back)
  * k_sat128r (k_sat128): DP lane-instructions = SQ_INSTS_VALU (wave-instructions) x 64 lanes x the double-precision share of the VALU
                 instructions in the sub-step loop (static census of the built object, isa_counts.json);
                 frac = DP lane-instructions / duration / (256 CUs x 4 SIMDs x 16 DP lanes x 2.4 GHz)
  * k_press128s: fp64 matrix flops = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512; frac = flops / duration / 78.6 TF
Durations: from the un-instrumented kernel trace (kernel_stats, AverageNs) when given -- counter collection serialises and
slows dispatches -- else from the counter file's own timestamps."""
import csv
import json
import re
import sys
from collections import defaultdict

def short(name):
    """k_sat128, k_nd_sub, k_nd_wave<7, 3, 4> ...: the kernel's name with its template arguments where several instantiations run."""
    m = re.search(r"(k_nd_(?:wave|top)<[^>]*>)", name)
    if m:
        return m.group(1)
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:40]


DP_LANE_RATE = 256 * 4 * 16 * 2.4e9
FP64_PEAK = 2 * DP_LANE_RATE

rows = list(csv.DictReader(open(sys.argv[1])))
stats = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        stats.setdefault(short(r["Name"]), float(r["AverageNs"]))
isa = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
bench_line = {}
if len(sys.argv) > 4:  # the JSON line bench.py printed under the kernel trace: measured mean sub-steps per member-step
    try:
        bench_line = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
    except Exception:
        bench_line = {}
disp = defaultdict(dict)
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"])
    disp[key][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[key]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    disp[key]["_grid"] = float(r["Grid_Size"])
    disp[key]["_wg"] = float(r["Workgroup_Size"])
agg = defaultdict(list)
for (_, name), c in disp.items():
    agg[short(name)].append(c)
out = {}
for name, cs in sorted(agg.items()):
    n = len(cs)
    avg = lambda k: sum(c.get(k, 0.0) for c in cs) / n  # noqa: E731
    ns_pmc = avg("_ns")
    ns = stats.get(name, ns_pmc)
    e = {"dispatches": n, "avg_us": ns / 1e3, "avg_us_under_pmc": ns_pmc / 1e3, "duration_source": "kernel trace" if name in stats else "pmc timestamps",
         "members_per_launch": avg("_grid") / max(avg("_wg"), 1.0),
         "SQ_INSTS_VALU": avg("SQ_INSTS_VALU"), "SQ_INSTS_VALU_MFMA_MOPS_F64": avg("SQ_INSTS_VALU_MFMA_MOPS_F64"),
         "SQ_VALU_MFMA_BUSY_CYCLES": avg("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CU_CYCLES": avg("SQ_BUSY_CU_CYCLES"),
         "GRBM_GUI_ACTIVE": avg("GRBM_GUI_ACTIVE"), "effective_clock_GHz_under_pmc": avg("GRBM_GUI_ACTIVE") / 8 / ns_pmc}
    lane_instr = avg("SQ_INSTS_VALU") * 64
    e["valu_lane_instr_per_launch"] = lane_instr
    if name in ("k_sat128", "k_sat128r") and name in isa:
        c = isa[name]["counts"]
        share = c["dp_valu"] / (c["dp_valu"] + c["other_valu"])
        e["dp_share_of_valu_from_isa_census"] = share
        e["dp_lane_instr_per_launch"] = lane_instr * share
        e["fp64_valu_frac_of_dp_lane_peak"] = lane_instr * share / (ns * 1e-9) / DP_LANE_RATE
        e["valu_issue_frac_if_every_instr_took_a_dp_slot"] = lane_instr / (ns * 1e-9) / DP_LANE_RATE
        nts = bench_line.get("roofline", {}).get("mean_nts")
        if nts:
            # instructions the sweep would execute if no wave skipped a dry band: (DP + other VALU per thread and sub-step) x 8 waves
            # x sub-steps x members; the measured count is lower by the dry-band skip
            algo = (c["dp_valu"] + c["other_valu"]) * 8.0 * nts * e["members_per_launch"]
            e["algorithmic_valu_wave_instr_per_launch"] = algo
            e["executed_over_algorithmic"] = avg("SQ_INSTS_VALU") / algo
            e["fp64_valu_frac_algorithmic_work"] = algo * 64 * share / (ns * 1e-9) / DP_LANE_RATE
    flops = avg("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512
    if flops > 0:
        e["fp64_mfma_flops_per_launch"] = flops
        e["fp64_mfma_frac_of_peak"] = flops / (ns * 1e-9) / FP64_PEAK
        e["mfma_busy_frac_of_cu_busy"] = avg("SQ_VALU_MFMA_BUSY_CYCLES") / max(avg("SQ_BUSY_CU_CYCLES"), 1.0)
    out[name] = e
# the nested-dissection pressure solve is several launches per time step: their matrix-core work and time together
nd = {k: v for k, v in out.items() if k.startswith("k_nd_")}
pressure_nd = None
if nd:
    # (a member per workgroup in k_nd_assemble and k_nd_flux; k_nd_top: round 5 runs level 4 as a launch of its own, a front per workgroup)
    members = max((v["members_per_launch"] for k, v in nd.items() if k in ("k_nd_assemble", "k_nd_flux")), default=None) if any(k.startswith("k_nd_top") for k in nd) else None
    flops = sum(v.get("fp64_mfma_flops_per_launch", 0.0) for v in nd.values())
    us = sum(v["avg_us"] for v in nd.values())
    if members:
        pressure_nd = {"kernels": sorted(nd), "sum_avg_us": us, "members_per_step": members, "fp64_mfma_flops_per_member_step": flops / members,
                       "fp64_mfma_frac_of_peak": flops / (us * 1e-6) / FP64_PEAK,
                       "valu_wave_instr_per_member_step": sum(v["SQ_INSTS_VALU"] for v in nd.values()) / members}
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent))
try:
    from obj_hash import object_hashes
    hashes = object_hashes()
except Exception:
    hashes = None
print(json.dumps({"object_sha256": hashes, "pressure_nd": pressure_nd, "peaks": {"dp_lane_instr_per_s": DP_LANE_RATE, "fp64_tflops": FP64_PEAK / 1e12, "assumed_clock_GHz": 2.4}, "kernels": out}, indent=1))
