# PMC passes over the RePaint kernels of the C3 window (separate rocprofv3 runs; bench.py's K2 measurement = 2 launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/k2pmc
mkdir -p $OUT
B="python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- $B > $OUT/b_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- $B > $OUT/b_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/sq -o s -- $B > $OUT/b_sq.json 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/sq2 -o s -- $B > $OUT/b_sq2.json 2> $OUT/sq2.err
for d in fetch write sq sq2; do python tools/pmc_kernel.py $OUT/$d repaint_ > $OUT/$d.txt 2>&1; done
rm -rf $OUT/fetch $OUT/write $OUT/sq $OUT/sq2
cat $OUT/fetch.txt $OUT/write.txt $OUT/sq.txt $OUT/sq2.txt
