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
python3 tests/tools/cpu_baselines_extra.py > $OUT/cpu_baselines_extra.txt 2> $W/extra.err
ls -la $OUT
