#!/bin/bash
# Collect the judged profile set on the GPU box (run from the repo root through gpurun):
#   bash profiles/tools/collect.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command          -> kernel_stats_*.csv
# 2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (256 members)       -> pmc_*_counter_collection.csv
# 3. pmc_to_json.py applies the gfx950 correction                           -> pmc_hbm_traffic.json
# 4. the same for the ensemble-smoother update (diag/bench_update.py)        -> kernel_stats_update.csv
set -u
R=${1:-r01}
OUT=gpurun_out/profiles/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/hmprof; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-esmda > $OUT/bench_under_rocprof.json 2> $W/ks.err
cp "$(find $W/ks -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench_steps2_warmup1.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- python3 bench.py --members 256 --steps 1 --warmup 0 --no-cpu-baseline --no-esmda > /dev/null 2> $W/$C.err
  cp "$(find $W/$C -name '*counter_collection.csv' | head -1)" $OUT/pmc_${C}_counter_collection.csv
done
python3 profiles/tools/pmc_to_json.py $OUT 256 > $OUT/pmc_hbm_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $W/upd -o upd -- python3 profiles/diag/bench_update.py > $OUT/bench_update.txt 2> $W/upd.err
cp "$(find $W/upd -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_update.csv
# 5. matrix-core counters of the update kernels (own pass, counters only)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/mf -o mf -- python3 profiles/diag/bench_update.py > /dev/null 2> $W/mf.err
f=$(find $W/mf -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
  (head -1 $f; grep -E "gxt|apply|dgemm|spd_inverse" $f) > $OUT/pmc_mfma_update_counter_collection.csv
  python3 profiles/tools/mfma_util.py $OUT/pmc_mfma_update_counter_collection.csv > $OUT/mfma_utilisation_update.json
else
  tail -5 $W/mf.err
fi
# the raw per-dispatch PMC files are large: keep only the rows of our kernels
for C in FETCH_SIZE WRITE_SIZE; do
  f=$OUT/pmc_${C}_counter_collection.csv
  (head -1 $f; grep -E "k_press|k_sat|k_perm|k_pressure|k_saturation" $f) > $f.tmp && mv $f.tmp $f
done
ls -la $OUT
