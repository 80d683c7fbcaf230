#!/bin/bash
# A/B copies of the library with press_nd.hip compiled under extra -D flags:  diag/build_nd_ab.sh name "-DX=1 -DY=0"  -> build_ab/libhm_<name>.so
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
out=../../build_ab
mkdir -p $out
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $2 -c press_nd.hip -o $out/press_nd_$1.o
objs=$(ls *.o | grep -v '^press_nd.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_$1.so $objs $out/press_nd_$1.o -lpthread -ldl
echo "built $out/libhm_$1.so"
