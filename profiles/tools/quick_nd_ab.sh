#!/bin/bash
# wall time of the pressure step for every build_ab/libhm_*.so (profiles/diag/build_nd_ab.sh, nd_time.py)
mkdir -p gpurun_out/r05b
for l in build_ab/libhm_*.so; do
  n=$(basename $l .so); n=${n#libhm_}
  echo "$n: $(HM_AMD_LIB=$l python profiles/diag/nd_time.py 1000 10 2>&1 | tail -2 | tr '\n' ' ')" | tee -a gpurun_out/r05b/nd_ab_${1:-x}.txt
done
