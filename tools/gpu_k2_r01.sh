# counters of the RePaint kernel (one window of a C3-like chunk)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/k2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/k2/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --skip-cpu --skip-alt > gpurun_out/k2/bench.json 2> gpurun_out/k2/sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/k2/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --skip-cpu --skip-alt > /dev/null 2> gpurun_out/k2/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/k2/write -o w -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --skip-cpu --skip-alt > /dev/null 2> gpurun_out/k2/w.err
