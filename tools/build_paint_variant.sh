#!/bin/bash
# tools/build_paint_variant.sh NAME "EXTRA FLAGS": relate_amd/variants/librelate_amd_NAME.so = the library with the
# Li-Stephens kernels (paint_kernels.hip, repaint_kernels.hip; all three summation modes) recompiled under EXTRA
# (e.g. "-DRL_STATS -DRL_ONLY_S=80"); tools/exp_stats.py and the K1 experiments load it through RELATE_AMD_LIB.
set -e
cd "$(dirname "$0")/../relate_amd/csrc"
NAME=$1; EXTRA=$2
mkdir -p ../variants ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. -Wno-unused-result"
pids=()
for m in 0 1 2; do
  for f in paint_kernels repaint_kernels; do
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -DRL_MODE=$m -c $f.hip -o ../../build/variants/${f}_m${m}_$NAME.o &
    pids+=($!)
  done
done
for p in "${pids[@]}"; do wait $p; done
OBJS=$(ls ../../build/obj/*.o | grep -v "paint_kernels_m")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librelate_amd_$NAME.so $OBJS ../../build/variants/*_$NAME.o -lpthread -lz
echo built ../variants/librelate_amd_$NAME.so
