#!/bin/bash
# A second copy of the library with press128s.hip compiled with extra flags, without touching the in-tree objects:
#   profiles/diag/build_press_prof.sh [name [flags]]   ->  build_prof/libhm_<name>.so   (remove build_prof/ afterwards)
# default: name = prof, flags = -DHM_PRESS_PROF  (the cycle stamps diag/press_prof.py reads)
set -e
cd "$(dirname "$0")/../../historymatching_amd/csrc"
name=${1:-prof}
flags=${2:--DHM_PRESS_PROF}
out=../../build_prof
mkdir -p $out
make -s
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $flags -c press128s.hip -o $out/press128s_$name.o
objs=$(ls *.o | grep -v '^press128s.o$')
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhm_$name.so $objs $out/press128s_$name.o -lpthread -ldl
echo "built $out/libhm_$name.so"
