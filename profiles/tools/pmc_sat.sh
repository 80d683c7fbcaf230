#!/bin/bash
# SQ counters of the fp64 saturation sweeps (k_sat128e default, k_sat128 = sat_variant 5), separate --pmc passes, counters only.
export TMPDIR=/tmp
W=/tmp/pmcsat; rm -rf $W; mkdir -p $W gpurun_out/pmc_sat
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $W/p$i -o p -- python3 tests/tools/sat_edge_ab.py 1000 6 > $W/p$i.out 2> $W/p$i.err
  f=$(find $W/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "k_sat128" $f) > gpurun_out/pmc_sat/p$i.csv; else tail -5 $W/p$i.err; fi
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_sat/p*.csv")):
    for r in csv.DictReader(open(f)):
        k = "e" if "sat128e" in r["Kernel_Name"] else "old"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}  last={v[-1]:.4g}")
PY
