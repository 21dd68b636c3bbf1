#!/bin/bash
# Register / LDS / spill report of the kernels of one source file (compile-only, no GPU needed):
#   tools/kernel_resources.sh conv_wino24 [extra hipcc flags]
F=$1; shift
cd "$(dirname "$0")/../orienmask_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $F.hip -o /tmp/kr_$F.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|Spill|LDS Size|Occupancy|SGPRs:|ScratchSize" | sed 's/.*remark: [^:]*:[0-9]*:[0-9]*: //' | paste - - - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{$1=$1};1'
