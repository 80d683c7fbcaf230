#!/bin/bash
# Round-6 profile set (run from the repo root through gpurun):   bash profiles/tools/collect_r06.sh r06
#  1. rocprofv3 --kernel-trace --stats of the bench command (forward legs only)             -> kernel_stats_bench.csv
#  2. own --pmc pass, counters only: fp64 VALU / matrix counters of the forward kernels      -> pmc_fp64_forward_counter_collection.csv
#  3. own --pmc FETCH_SIZE / WRITE_SIZE passes (256 members)                                -> pmc_hbm_traffic.json
#  4. isa_counts.json + fp64_roofline.json, both carrying the sha256 of the objects they were taken from
#  5. the larger grids' nested dissection: per-level launch times at 256 x 256 / 512 members and 512 x 512 / 125 members
#     (ndl_levels_c4.txt, ndl_levels_c5.txt), the direct solver beside the two-level CG (pressure_variants_large.txt), kernel statistics of
#     whole shards (kernel_stats_large_grid_c4/c5.csv), FETCH / WRITE passes of the pressure step at 256 x 256 (pmc_hbm_traffic_nd256.json)
#  6. the analysis step in situ against back to back (upd_in_situ.txt), update kernel statistics (kernel_stats_update.csv)
#  7. python3 bench.py -> bench_default.json (the default line of this build)
set -u
R=${1:-r06}
OUT=gpurun_out/profiles/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/hmprof; rm -rf $W; mkdir -p $W
LEGS="--no-cpu-baseline --no-esmda --no-config4 --no-config5 --no-two-streams --no-host-call --no-strong-shard"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python3 bench.py --steps 2 --warmup 1 $LEGS > $OUT/bench_under_rocprof.json 2> $W/ks.err
cp "$(find $W/ks -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/f64 -o f64 -- python3 bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2> $W/f64.err
f=$(find $W/f64 -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then (head -1 $f; grep -E "k_nd_|k_press|k_sat" $f) > $OUT/pmc_fp64_forward_counter_collection.csv; else tail -5 $W/f64.err; fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- python3 bench.py --members 256 --steps 1 --warmup 0 $LEGS > /dev/null 2> $W/$C.err
  f=$(find $W/$C -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep -E "k_nd_|k_press|k_sat|k_perm|k_pressure|k_saturation" $f) > $OUT/pmc_${C}_counter_collection.csv
done
python3 profiles/tools/pmc_to_json.py $OUT 256 > $OUT/pmc_hbm_traffic.json
python3 - <<PY > $OUT/isa_counts.json
import json, subprocess, sys
sys.path.insert(0, "profiles/tools")
from obj_hash import object_hashes
def census(obj, sub):
    return json.loads(subprocess.run(["python3", "profiles/tools/isa_count.py", obj, sub, "32", "2"], capture_output=True, text=True, check=True).stdout)
print(json.dumps({"object_sha256": object_hashes(), "k_sat128r": census("historymatching_amd/csrc/sat128r.o", "k_sat128rILb1"),
                  "k_sat128": census("historymatching_amd/csrc/sat128.o", "k_sat128ILb1"),
                  "how": "profiles/tools/isa_count.py historymatching_amd/csrc/sat128r.o k_sat128rILb1 32 2 (and sat128.o k_sat128ILb1: sat_variant 5)"}, indent=1))
PY
python3 profiles/tools/fp64_roofline.py $OUT/pmc_fp64_forward_counter_collection.csv $OUT/kernel_stats_bench.csv $OUT/isa_counts.json $OUT/bench_under_rocprof.json > $OUT/fp64_roofline.json
# 5. the larger grids
for cfg in "256 512 c4" "512 125 c5"; do
  set -- $cfg
  rm -rf $W/ndl; rocprofv3 --kernel-trace --output-format csv -d $W/ndl -o t -- python3 tests/tools/ndl_check.py $1 $2 0 > $OUT/pressure_variants_large_$3.txt 2> $W/ndl.err
  python3 profiles/tools/ndl_levels.py $W/ndl $2 > $OUT/ndl_levels_$3.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/$3 -o ks -- python3 tests/tools/large_grid_timing.py $1 $2 39 > $OUT/large_grid_$3.txt 2> $W/$3.err
  cp "$(find $W/$3 -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_large_grid_$3.csv
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/pl$C -o pmc -- python3 tests/tools/large_grid_timing.py 256 128 1 > /dev/null 2> $W/pl$C.err
  f=$(find $W/pl$C -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep -E "k_nd_|k_big_|k_ndl_|k_sat256s" $f) > $OUT/pmc_nd256_$C.csv
done
python3 profiles/tools/pmc_nd_large_to_json.py $OUT 128 > $OUT/pmc_hbm_traffic_nd256.json
python3 tests/tools/nd_residual_stats.py 256 1024 1000 > $OUT/nd_residual_stats.txt 2>&1
# 6. the analysis step
rm -rf $W/upd; rocprofv3 --kernel-trace --stats --output-format csv -d $W/upd -o upd -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config4 --no-config5 --no-two-streams --no-host-call --no-strong-shard > $OUT/bench_update_legs.json 2> $W/upd.err
cp "$(find $W/upd -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_update.csv
python3 profiles/tools/upd_in_situ.py $W/upd > $OUT/upd_in_situ.txt
# 6b. round 6: small member shards (one rank's share of a strong-scaled config 2), config 1, the fp32 mode's timing, the fp32 shards' kernel statistics
python3 profiles/diag/small_shards.py 64,32 64,125,250,500,1000 > $OUT/small_shards.txt 2>&1
python3 tests/tools/config1_timing.py 100 20 > $OUT/config1_timing.txt 2>&1
python3 tests/tools/fp32_mode_timing.py 1000 > $OUT/fp32_mode_timing_128.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $W/c5f -o ks -- python3 tests/tools/large_grid_timing.py 512 125 39 0 32 > $OUT/large_grid_c5_fp32.txt 2> $W/c5f.err
cp "$(find $W/c5f -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_large_grid_c5_fp32.csv
python3 tests/tools/large_grid_timing.py 256 512 39 0 32 > $OUT/large_grid_c4_fp32.txt 2>&1
python3 profiles/tools/nd_kernel_rooflines.py $OUT/pmc_hbm_traffic.json $OUT/kernel_stats_bench.csv > $OUT/nd_kernel_rooflines.txt 2>&1
# 7. the default bench line of this build (it prices its launch times with the newest committed counts under profiles/rNN/: this round's go there first,
#    so that the line's `stale_inputs` refers to the objects that ran)
mkdir -p profiles/$R
cp $OUT/isa_counts.json $OUT/fp64_roofline.json $OUT/pmc_hbm_traffic.json profiles/$R/
python3 bench.py > $OUT/bench_default.json 2> $W/bench.err
ls -la $OUT
