#!/bin/bash
# A/B copies of the library with ONE source compiled under extra -D flags:  diag/build_src_ab.sh sat256s name "-DX=1"  -> build_ab/libhm_<name>.so
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
out=../../build_ab
mkdir -p $out
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $3 -c $1.hip -o $out/$1_$2.o
objs=$(ls *.o | grep -v "^$1.o\$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_$2.so $objs $out/$1_$2.o -lpthread -ldl
echo "built $out/libhm_$2.so"
