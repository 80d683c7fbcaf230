#!/bin/bash
# Round 5: the matrix-core counter pass of the analysis step (own run, counters only): SQ_INSTS_VALU_MFMA_MOPS_F32/F64 etc. at config 3's shape
#   -> gpurun_out/profiles/r05/pmc_mfma_update_counter_collection.csv, mfma_utilisation_update.json
export TMPDIR=/tmp
OUT=gpurun_out/profiles/r05; mkdir -p $OUT
W=/tmp/hmprof_upd; rm -rf $W; mkdir -p $W
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/mf -o mf -- python3 profiles/diag/bench_update.py > /dev/null 2> $W/mf.err
f=$(find $W/mf -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
  (head -1 $f; grep -E "k_gxt|k_apply|k_ldl|k_center|k_spd|k_dgemm|k_gram" $f) > $OUT/pmc_mfma_update_counter_collection.csv
  python3 profiles/tools/mfma_util.py $OUT/pmc_mfma_update_counter_collection.csv > $OUT/mfma_utilisation_update.json
  python3 -c "
import json
d=json.load(open('$OUT/mfma_utilisation_update.json'))
for k,v in d.items(): print(k[:50], v['dispatches'], round(v['avg_us_under_pmc'],1), 'us', round(v['frac_of_matrix_peak'],3))"
else tail -5 $W/mf.err; fi
