# PMC passes (HBM traffic) of the bench command, one counter set per pass
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk > gpurun_out/pmc/fetch_bench.json 2> gpurun_out/pmc/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/write -o write -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk > gpurun_out/pmc/write_bench.json 2> gpurun_out/pmc/write.err
ls -la gpurun_out/pmc/*
