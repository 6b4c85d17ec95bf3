# scalar-memory latency counters of the Paint kernels
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/sq2
for mode in lanes; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH --kernel-trace -d gpurun_out/sq2/$mode -o sq -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --mode $mode --skip-cpu --skip-alt --skip-k23 > gpurun_out/sq2/${mode}_bench.json 2> gpurun_out/sq2/$mode.err
done
