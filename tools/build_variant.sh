#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS": relate_amd/variants/librelate_amd_NAME.so = the library with
# minmatch_gpu.hip recompiled under EXTRA (experiments on the tree-build kernel, tools/bench_builder_variants.py)
set -e
cd "$(dirname "$0")/../relate_amd/csrc"
NAME=$1; EXTRA=$2
mkdir -p ../variants ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c minmatch_gpu.hip -o ../../build/variants/minmatch_gpu_$NAME.o
OBJS=$(ls ../../build/obj/*.o | grep -v minmatch_gpu.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librelate_amd_$NAME.so $OBJS ../../build/variants/minmatch_gpu_$NAME.o -lpthread -lz
echo built ../variants/librelate_amd_$NAME.so
