#!/bin/bash
# Round-2 profile set (run from the repo root through gpurun):   bash profiles/tools/collect_r02.sh r02
#  1. rocprofv3 --kernel-trace --stats of the bench command (forward legs only)            -> kernel_stats_bench.csv
#  2. own --pmc pass, counters only, fp64 VALU / matrix counters of the forward kernels     -> pmc_fp64_forward_counter_collection.csv
#  3. own --pmc FETCH_SIZE / WRITE_SIZE passes (256 members)                               -> pmc_hbm_traffic.json
#  4. the update at config 3's shape: kernel trace + matrix-core counters                   -> kernel_stats_update.csv, mfma_utilisation_update.json
#  5. fp64_roofline.py: the roofline fractions bench.py prints, from the CSVs               -> fp64_roofline.json
set -u
R=${1:-r02}
OUT=gpurun_out/profiles/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/hmprof; rm -rf $W; mkdir -p $W
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python3 $BENCH > $OUT/bench_under_rocprof.json 2> $W/ks.err
cp "$(find $W/ks -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/f64 -o f64 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams > /dev/null 2> $W/f64.err
f=$(find $W/f64 -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
  (head -1 $f; grep -E "k_press|k_sat" $f) > $OUT/pmc_fp64_forward_counter_collection.csv
else
  tail -5 $W/f64.err
fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- python3 bench.py --members 256 --steps 1 --warmup 0 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams > /dev/null 2> $W/$C.err
  f=$(find $W/$C -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep -E "k_press|k_sat|k_perm|k_pressure|k_saturation" $f) > $OUT/pmc_${C}_counter_collection.csv
done
python3 profiles/tools/pmc_to_json.py $OUT 256 > $OUT/pmc_hbm_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $W/upd -o upd -- python3 profiles/diag/bench_update.py > $OUT/bench_update.txt 2> $W/upd.err
cp "$(find $W/upd -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_update.csv
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/mf -o mf -- python3 profiles/diag/bench_update.py > /dev/null 2> $W/mf.err
f=$(find $W/mf -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
  (head -1 $f; grep -E "gxt|apply|dgemm|spd_inverse|ldl_|center_gram|k_upd" $f) > $OUT/pmc_mfma_update_counter_collection.csv
  python3 profiles/tools/mfma_util.py $OUT/pmc_mfma_update_counter_collection.csv > $OUT/mfma_utilisation_update.json
else
  tail -5 $W/mf.err
fi
python3 - <<PY > $OUT/isa_counts.json
import json, subprocess
d = json.loads(subprocess.run(["python3", "profiles/tools/isa_count.py", "historymatching_amd/csrc/sat128.o", "k_sat128ILb1", "32", "2"], capture_output=True, text=True, check=True).stdout)
print(json.dumps({"k_sat128": d, "how": "profiles/tools/isa_count.py historymatching_amd/csrc/sat128.o k_sat128ILb1 32 2"}, indent=1))
PY
python3 profiles/tools/fp64_roofline.py $OUT/pmc_fp64_forward_counter_collection.csv $OUT/kernel_stats_bench.csv $OUT/isa_counts.json $OUT/bench_under_rocprof.json > $OUT/fp64_roofline.json
ls -la $OUT
