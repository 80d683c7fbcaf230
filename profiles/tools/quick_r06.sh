#!/bin/bash
# round 6: per-kernel times of the 128 x 128 pressure step for one build of the library (kernel trace of diag/nd_time.py ... run: three 40-step runs with the dry-front
# reuse, 120 launches of every kernel), and the build's bits (diag/nd_bits.py <tag>)
#     profiles/tools/quick_r06.sh <tag> [lib]
tag=${1:-x}; lib=${2:-historymatching_amd/libhm_amd.so}
mkdir -p gpurun_out/r06/$tag
case $lib in /*) ;; *) lib=$GRAFT_REPO_ROOT/$lib;; esac
export HM_AMD_LIB=$lib
python profiles/diag/nd_bits.py $tag > gpurun_out/r06/$tag/nd_bits.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06/$tag/trace -o nd -- python3 $GRAFT_REPO_ROOT/profiles/diag/nd_time.py 1000 10 run > $GRAFT_REPO_ROOT/gpurun_out/r06/$tag/nd_time.txt 2> /dev/null || exit 1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r06/$tag/trace -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r06/$tag/kernel_stats_nd_time.csv
rm -rf gpurun_out/r06/$tag/trace
python profiles/tools/print_stats.py gpurun_out/r06/$tag/kernel_stats_nd_time.csv 2>/dev/null | head -30
cat gpurun_out/r06/$tag/nd_time.txt
python profiles/diag/nd_time.py 1000 10 all
