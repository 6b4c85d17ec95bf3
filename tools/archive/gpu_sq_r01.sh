# SQ issue/stall breakdown of the Paint kernels (one PMC pass; 8 SQ slots + GRBM)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/sq
for mode in exact lanes; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/sq/$mode -o sq -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --mode $mode --skip-cpu --skip-alt --skip-k23 > gpurun_out/sq/${mode}_bench.json 2> gpurun_out/sq/$mode.err
done
ls -la gpurun_out/sq/*
