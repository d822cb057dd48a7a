#!/bin/bash
# register / spill report of the kernels whose name matches $1 (compiles pse_kernels.hip or $2 to /tmp)
cd "$(dirname "$0")/../../pse_amd/csrc" || exit 1
f=${2:-pse_kernels.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|$1" -A12 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//; s/EEvP15HIP.*//'
