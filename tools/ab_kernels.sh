#!/bin/bash
# A variant of the library that differs only in pc_kernels.hip:  tools/ab_kernels.sh <name> [-DPC_...]
# (the other objects are the main build's: run `make -C porechop_amd/csrc` first)
# -> porechop_amd/libporechop_amd_<name>.so  (use with PC_LIBRARY=$PWD/porechop_amd/libporechop_amd_<name>.so)
NAME=$1; shift
cd "$(dirname "$0")/../porechop_amd/csrc"
D=/tmp/abk_$NAME; mkdir -p $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c pc_kernels.hip -o $D/pc_kernels.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libporechop_amd_$NAME.so $D/pc_kernels.o pc_reduce.o pc_prefilter.o pc_select.o pc_middle.o pc_slow.o pc_api.o pc_jit.o pc_io.o -ldl -lz -lpthread && echo built ../libporechop_amd_$NAME.so
