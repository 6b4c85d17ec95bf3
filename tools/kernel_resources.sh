#!/bin/bash
# tools/kernel_resources.sh FILE.hip "EXTRA FLAGS": registers, scratch and LDS of every kernel of FILE as the compiler
# reports them (-Rpass-analysis=kernel-resource-usage), one line per kernel.
cd "$(dirname "$0")/../relate_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. \
  -Wno-unused-result $2 -Rpass-analysis=kernel-resource-usage --cuda-device-only -c $1 -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size|Spill" |
  sed -e 's/.*remark: [^ ]* //' | paste - - - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | tr -s ' \t' ' '
