#!/bin/bash
# Per-kernel resource usage (VGPRs, SGPRs, scratch, LDS, occupancy) of one csrc/*.hip file, and its gfx950 assembly in /tmp/kres/.
#   tools/kres.sh surf_kernels [name-filter]
F=${1:-surf_kernels}; PAT=${2:-.}
mkdir -p /tmp/kres
cd "$(dirname "$0")/../imagestitch_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $F.hip -o /tmp/kres/$F.o 2>&1 |
  grep -E "error|warning: |Name:|SGPRs:|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' |
  awk '/Function Name|Name:/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | grep -E "$PAT"
