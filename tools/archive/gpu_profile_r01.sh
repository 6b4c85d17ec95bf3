# Round-1 evidence run on the GPU box: tests, bench, rocprofv3 kernel stats, PMC passes.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r01/pytest_gpu.txt
python bench.py > gpurun_out/r01/bench_c3.json 2> gpurun_out/r01/bench_c3.err
python bench.py --haplotypes 1000 --snps 100000 --memory 5 > gpurun_out/r01/bench_c2.json 2> gpurun_out/r01/bench_c2.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r01/stats -o c3 -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --skip-chunk > gpurun_out/r01/bench_under_rocprof.json 2> gpurun_out/r01/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r01/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk > gpurun_out/r01/bench_fetch.json 2> gpurun_out/r01/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r01/write -o write -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk > gpurun_out/r01/bench_write.json 2> gpurun_out/r01/write.err
cat gpurun_out/r01/pytest_gpu.txt gpurun_out/r01/bench_c3.json
