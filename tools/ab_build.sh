#!/bin/bash
# Build a variant of the library next to the real one:  tools/ab_build.sh <name> [extra hipcc flags, e.g. -DPC_VARIANT_X=1]
# -> porechop_amd/libporechop_amd_<name>.so  (use with PC_LIBRARY=$PWD/porechop_amd/libporechop_amd_<name>.so)
NAME=$1; shift
cd "$(dirname "$0")/../porechop_amd/csrc"
D=/tmp/ab_$NAME; mkdir -p $D
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@"
for f in pc_kernels pc_reduce pc_prefilter pc_select pc_slow; do hipcc $F -c $f.hip -o $D/$f.o & done
hipcc $F -x hip -c pc_api.cpp -o $D/pc_api.o &
hipcc $F -x hip -c pc_jit.cpp -o $D/pc_jit.o &
g++ -O3 -std=c++17 -fPIC -Wall -pthread -c pc_io.cpp -o $D/pc_io.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libporechop_amd_$NAME.so $D/*.o -ldl -lz -lpthread && echo built ../libporechop_amd_$NAME.so
