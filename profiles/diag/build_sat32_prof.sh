#!/bin/bash
# A second copy of the library with sat32s.hip compiled with cycle stamps (-DHM_SAT_PROF=<workgroup id>):  -> build_prof/libhm_sat32prof.so
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
out=../../build_prof
mkdir -p $out
make -s
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DHM_SAT_PROF=${1:-0} ${2:-} -c sat32s.hip -o $out/sat32s_prof.o
objs=$(ls *.o | grep -v '^sat32s.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_sat32prof.so $objs $out/sat32s_prof.o -lpthread -ldl
echo "built $out/libhm_sat32prof.so"
