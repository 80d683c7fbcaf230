#!/bin/bash
# Round-3 profile set (run from the repo root through gpurun):   bash profiles/tools/collect_r03.sh r03
#  1. rocprofv3 --kernel-trace --stats of the bench command (forward legs only)             -> kernel_stats_bench.csv
#  2. own --pmc pass, counters only: fp64 VALU / matrix counters of the forward kernels      -> pmc_fp64_forward_counter_collection.csv
#  3. own --pmc FETCH_SIZE / WRITE_SIZE passes (256 members)                                -> pmc_hbm_traffic.json
#  4. the update at config 3's shape: kernel trace                                          -> kernel_stats_update.csv
#  5. isa_counts.json + fp64_roofline.json, both carrying the sha256 of the objects they were taken from
#  6. the pressure variants side by side (nested dissection vs block elimination)            -> pressure_variants.txt
#     the saturation sweeps side by side (fw image in LDS vs fw in registers)               -> saturation_variants.txt
#  7. pressure step alone: kernel_stats_nd_only.txt, occupancy_nd.txt, nd_reuse.txt;  8. python3 bench.py -> bench_default.json
set -u
R=${1:-r03}
OUT=gpurun_out/profiles/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/hmprof; rm -rf $W; mkdir -p $W
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams --no-host-call"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/ks -o ks -- python3 $BENCH > $OUT/bench_under_rocprof.json 2> $W/ks.err
cp "$(find $W/ks -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/f64 -o f64 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams --no-host-call > /dev/null 2> $W/f64.err
f=$(find $W/f64 -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
  (head -1 $f; grep -E "k_nd_|k_press|k_sat" $f) > $OUT/pmc_fp64_forward_counter_collection.csv
else
  tail -5 $W/f64.err
fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- python3 bench.py --members 256 --steps 1 --warmup 0 --no-cpu-baseline --no-esmda --no-config4 --no-two-streams --no-host-call > /dev/null 2> $W/$C.err
  f=$(find $W/$C -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep -E "k_nd_|k_press|k_sat|k_perm|k_pressure|k_saturation" $f) > $OUT/pmc_${C}_counter_collection.csv
done
python3 profiles/tools/pmc_to_json.py $OUT 256 > $OUT/pmc_hbm_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $W/upd -o upd -- python3 profiles/diag/bench_update.py > $OUT/bench_update.txt 2> $W/upd.err
cp "$(find $W/upd -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_update.csv
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
python3 tests/tools/nd_check.py 1000 20 > $OUT/pressure_variants.txt 2>&1
python3 tests/tools/sat_check.py 1000 5,0 > $OUT/saturation_variants.txt 2>&1
python3 tests/tools/long_parity.py 4 > $OUT/long_parity.txt 2>&1
python3 tests/tools/ies_iterate_timing.py > $OUT/ies_iterate.txt 2>&1
# 7. the pressure step alone: kernel trace of whole runs, occupancy counters (every front eliminated), the reuse of dry fronts
rocprofv3 --kernel-trace --stats --output-format csv -d $W/nd -o nd -- python3 tests/tools/nd_time.py 12 1000 1 > $W/nd.out 2> $W/nd.err
python3 profiles/tools/print_stats.py $W/nd > $OUT/kernel_stats_nd_only.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/occ -o occ -- python3 tests/tools/nd_time.py 14 1000 1 > /dev/null 2> $W/occ.err
python3 profiles/tools/occupancy.py "$(find $W/occ -name '*counter_collection.csv' | head -1)" k_nd_ > $OUT/occupancy_nd.txt
(python3 tests/tools/nd_reuse_check.py 1000 8 40; python3 tests/tools/dry_fraction.py 8) > $OUT/nd_reuse.txt 2>&1
# 8. the default bench line of this build
python3 bench.py > $OUT/bench_default.json 2> $W/bench.err
ls -la $OUT
