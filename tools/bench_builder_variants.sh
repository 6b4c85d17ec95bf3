#!/bin/bash
# On the GPU box: dump the matrices of builds 3..5 of section 0 of the N=5000 x L=20000 chunk, then time every
# library under relate_amd/variants (and the default one) on them.
D=/tmp/mmdump; rm -rf $D; mkdir -p $D
RELATE_AMD_TEST_MM_DUMP=$D:3:5 RELATE_AMD_GPU_BUILD=1 timeout 300 python tools/chunk_wallclock_big.py 5000 20000 20 1 > /dev/null 2>&1
ls -la $D | head
python tools/bench_builder_one.py $D
for so in relate_amd/variants/*.so; do
  [ -f "$so" ] && RELATE_AMD_LIB=$PWD/$so python tools/bench_builder_one.py $D
done
