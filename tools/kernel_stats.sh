#!/bin/bash
# Per-kernel VGPR / occupancy / code size of the ahead-of-time kernels (no GPU needed).
# tools/kernel_stats.sh [out.s]
S=${1:-/tmp/pc_kernels.s}
cd "$(dirname "$0")/../porechop_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o $S pc_kernels.hip 2>/dev/null
awk '/^_Z[A-Za-z0-9_]+:/{name=$1} /; NumVgprs:/{v=$3} /; ScratchSize:/{sc=$3} /; codeLenInByte/{c=$4} /; Occupancy:/{o=$3; printf "%-70s vgpr=%s scratch=%s occ=%s code=%s\n", name, v, sc, o, c}' $S | c++filt | sed 's/(pck::ScanArgs)//;s/void pck:://'
