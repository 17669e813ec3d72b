#!/bin/bash
# Register / spill / occupancy figures of every kernel variant, from the compiler's own resource report:
#   tools/kernel_regs.sh [render_kernels.hip|path_kernels.hip|bvh_gpu.hip] ["-DEXTRA=1 ..."]
cd "$(dirname "$0")/../directx-raytracer_amd/csrc"
SRC=${1:-render_kernels.hip}
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $2 -Rpass-analysis=kernel-resource-usage -c $SRC -o /tmp/kernel_regs.o 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/ { if (line) print line; line = $0; next } { line = line " | " $0 } END { print line }' | while read -r l; do
    n=$(echo "$l" | sed -E 's/Function Name: ([^ ]+).*/\1/' | c++filt | sed -E 's/crt::\(anonymous namespace\):://g; s/\(crt::RenderParams\)//')
    echo "$n | ${l#* | }"
  done
