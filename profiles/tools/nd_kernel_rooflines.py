"""Per-kernel HBM roofline of the 128 x 128 nested-dissection pressure step: measured HBM bytes per member (pmc_hbm_traffic.json: FETCH x 2 +
WRITE, the guide's correction calibrated on k_nd_solve's factor stream) x 1000 members over the kernel's average launch time in the bench run
(kernel_stats_bench.csv), against the 8 TB/s peak and the 6.3 TB/s the guide calls achievable.
     python3 profiles/tools/nd_kernel_rooflines.py <pmc_hbm_traffic.json> <kernel_stats_bench.csv> [members=1000]"""
import csv
import json
import sys

pmc = json.load(open(sys.argv[1]))["kernels"]
members = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
NAMES = {"nd_assemble": "k_nd_assemble", "nd_leaf": "k_nd_leaf(", "nd_sub": "k_nd_sub(", "nd_wave7": "k_nd_wave<7", "nd_wave6": "k_nd_wave<6", "nd_wave5": "k_nd_wave<5",
         "nd_top4": "k_nd_top<3, 5", "nd_top": "k_nd_top<3, 4, 13, 3, 16, false>", "nd_solve": "k_nd_solve(", "nd_solve_sub": "k_nd_solve_sub", "nd_leaf_solve": "k_nd_leaf_solve", "nd_flux": "k_nd_flux"}
avg = {}
for r in csv.DictReader(open(sys.argv[2])):
    for key, pat in NAMES.items():
        if pat in r["Name"]:
            avg[key] = float(r["AverageNs"]) * 1e-9
print(f"{'kernel':16s} {'MB / member':>12s} {'ms / launch':>12s} {'TB/s':>7s} {'of 8 TB/s':>10s} {'of 6.3':>8s}   (per launch of {members} members; bytes: FETCH x 2 + WRITE)")
tot_b = tot_t = 0.0
for key in NAMES:
    if key not in pmc or key not in avg:
        continue
    b = pmc[key]["hbm_bytes_per_member_corrected"] * members
    t = avg[key]
    tot_b += b
    tot_t += t
    print(f"{key:16s} {b / members / 1e6:12.2f} {t * 1e3:12.3f} {b / t / 1e12:7.2f} {b / t / 8e12:10.2f} {b / t / 6.3e12:8.2f}")
print(f"{'pressure step':16s} {tot_b / members / 1e6:12.2f} {tot_t * 1e3:12.3f} {tot_b / tot_t / 1e12:7.2f} {tot_b / tot_t / 8e12:10.2f} {tot_b / tot_t / 6.3e12:8.2f}")
