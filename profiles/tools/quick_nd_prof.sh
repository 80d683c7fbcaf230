#!/bin/bash
# cycle stamps of k_nd_top for waves 0, 1, 5 of block 0 (build_prof/libhm_ndprof{,1,5}.so: -DHM_ND_PROF_TOP_WAVE)
mkdir -p gpurun_out/r05b
for k in "" 1 5; do
  HM_AMD_LIB=build_prof/libhm_ndprof$k.so python profiles/diag/nd_prof.py 1000 > gpurun_out/r05b/nd_prof_${1:-x}_w${k:-0}.txt 2>&1
  head -16 gpurun_out/r05b/nd_prof_${1:-x}_w${k:-0}.txt
done
