#!/bin/bash
# tools/isa.sh <file.hip> <kernel mangled-name regex> : compile one translation unit with -save-temps and print the kernel's gfx950 ISA + its resources
set -e
cd "$(dirname "$0")/../imagestitch_amd/csrc"
F=$1; K=$2
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value $EXTRA -c $F -o build/isa/${F%.hip}.o -save-temps=obj
S=build/isa/${F%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
awk "/^${K}.*:/,/s_endpgm/" $S > /tmp/isa_out.s
wc -l /tmp/isa_out.s
grep -A45 "\.name: *${K}" $S | grep "vgpr_count\|sgpr_count\|group_segment\|private_segment\|vgpr_spill" || true
