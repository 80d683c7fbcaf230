#!/bin/bash
# A second copy of the library with sat128r.hip compiled with cycle stamps (-DHM_SAT_PROF):  -> build_prof/libhm_satprof.so
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
out=../../build_prof
mkdir -p $out
make -s
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DHM_SAT_PROF ${1:-} -c sat128r.hip -o $out/sat128r_prof.o
objs=$(ls *.o | grep -v '^sat128r.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_satprof.so $objs $out/sat128r_prof.o -lpthread -ldl
echo "built $out/libhm_satprof.so"
