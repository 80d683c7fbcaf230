#!/bin/bash
mkdir -p gpurun_out/r05b
HM_AMD_LIB=build_ab/libhm_t4.so python profiles/diag/nd_bits.py t4 2>&1 | tail -2
python profiles/diag/nd_bits.py new 2>&1 | tail -2
python profiles/diag/nd_bits.py t4 new
echo "t4: $(HM_AMD_LIB=build_ab/libhm_t4.so python profiles/diag/nd_time.py 1000 10 2>&1 | tail -2 | tr '\n' ' ')"
echo "new: $(python profiles/diag/nd_time.py 1000 10 2>&1 | tail -2 | tr '\n' ' ')"
