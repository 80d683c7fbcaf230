#!/bin/bash
for i in 1 2; do
echo "w16: $(HM_AMD_LIB=build_ab/libhm_w16.so python tests/tools/large_grid_timing.py 256 512 39 2>&1 | tail -1)"
echo "w8:  $(python tests/tools/large_grid_timing.py 256 512 39 2>&1 | tail -1)"
done
