#!/bin/bash
# Kernel statistics of the large-grid path on per-GPU shards of BASELINE configs 4 and 5 (run through gpurun):
#   bash profiles/tools/collect_large.sh r01
set -u
R=${1:-r01}
OUT=gpurun_out/profiles/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/hmprofL; rm -rf $W; mkdir -p $W
for cfg in "256 512 1 c4" "512 125 1 c5"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/$4 -o ks -- python3 tests/tools/large_grid_timing.py $1 $2 $3 > $OUT/large_grid_$4.txt 2> $W/$4.err
  cp "$(find $W/$4 -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_large_grid_$4.csv
done
# HBM traffic of the large-grid kernels: separate PMC passes on a smaller shard (128 members of 256 x 256)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $W/pl$C -o pmc -- python3 tests/tools/large_grid_timing.py 256 128 1 > /dev/null 2> $W/pl$C.err
  f=$(find $W/pl$C -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep -E "k_coarse_solve|k_tg_|k_sat128t|k_sat256s|k_press128s|k_tl_" $f) > $OUT/pmc_large_$C.csv
done
python3 profiles/tools/pmc_large_to_json.py $OUT 128 256 256 > $OUT/pmc_hbm_traffic_large.json
# the slab sweep (default at 256 wide) beside the tile teams, early in a run and at step 16
(for sv in 0 5; do python3 tests/tools/large_grid_timing.py 256 512 3 0 64 $sv; python3 tests/tools/large_grid_timing.py 256 512 16 0 64 $sv; done) > $OUT/large_grid_c4_slabs_vs_tile_teams.txt 2>&1
python3 tests/tools/cpu_baselines_extra.py > $OUT/cpu_baselines_extra.txt 2> $W/extra.err
ls -la $OUT
