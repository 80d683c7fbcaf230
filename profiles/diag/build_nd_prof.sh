#!/bin/bash
# A second copy of the library with press_nd.hip compiled with cycle stamps (-DHM_ND_PROF):  -> build_prof/libhm_ndprof.so
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
out=../../build_prof
mkdir -p $out
make -s
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DHM_ND_PROF ${1:-} -c press_nd.hip -o $out/press_nd_prof.o
objs=$(ls *.o | grep -v '^press_nd.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_ndprof.so $objs $out/press_nd_prof.o -lpthread -ldl
echo "built $out/libhm_ndprof.so"
