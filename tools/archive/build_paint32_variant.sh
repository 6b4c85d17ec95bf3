#!/bin/bash
# tools/build_paint32_variant.sh NAME "EXTRA FLAGS": relate_amd/variants/librelate_amd_NAME.so = the library with
# paint32_kernels.hip (the lanes32 kernel) recompiled under EXTRA, e.g. "-DRL_ONLY_S=80 -DRL32_WAVES_PER_SIMD=3"
set -e
cd "$(dirname "$0")/../relate_amd/csrc"
NAME=$1; EXTRA=$2
mkdir -p ../variants ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c paint32_kernels.hip -o ../../build/variants/paint32_$NAME.o
OBJS=$(ls ../../build/obj/*.o | grep -v "paint32_kernels")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librelate_amd_$NAME.so $OBJS ../../build/variants/paint32_$NAME.o -lpthread -lz
echo built ../variants/librelate_amd_$NAME.so
