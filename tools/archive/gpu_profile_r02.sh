# Round-2 evidence run on the GPU box: tests, bench, rocprofv3 kernel stats, PMC passes (separate runs).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r02/pytest_gpu.txt
python bench.py > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/stats -o c3 -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --skip-chunk > gpurun_out/r02/bench_under_rocprof.json 2> gpurun_out/r02/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > gpurun_out/r02/bench_fetch.json 2> gpurun_out/r02/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r02/write -o write -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > gpurun_out/r02/bench_write.json 2> gpurun_out/r02/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/r02/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt --snps 100000 > gpurun_out/r02/bench_sq.json 2> gpurun_out/r02/sq.err
cat gpurun_out/r02/pytest_gpu.txt gpurun_out/r02/bench_c3.json
