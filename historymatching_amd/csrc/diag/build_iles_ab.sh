#!/bin/bash
# A/B copies of the library with iles.hip compiled under extra -D flags:  diag/build_iles_ab.sh name "-DX=1"  -> build_ab/libhm_<name>.so
set -e
cd "$(dirname "$0")/.."
out=../../build_ab
mkdir -p $out
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $2 -c iles.hip -o $out/iles_$1.o
objs=$(ls *.o | grep -v '^iles.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_$1.so $objs $out/iles_$1.o -lpthread -ldl
echo "built $out/libhm_$1.so"
