# Round-2 final evidence on the GPU box: smoke, the -m gpu suite, bench (default run), rocprofv3 kernel stats of the bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02final
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o c3 -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --skip-chunk > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
python tools/rocprof_summary.py $(find $OUT/stats -name "*results.db" | head -1) > $OUT/kernel_stats_c3.txt 2>&1
rm -rf $OUT/stats
cat $OUT/smoke.txt | tail -2; cat $OUT/pytest_gpu.txt; head -16 $OUT/kernel_stats_c3.txt
