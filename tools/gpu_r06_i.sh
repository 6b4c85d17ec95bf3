# Round 6, run I (final): smoke, the whole GPU suite, the default bench line, the rocprofv3 kernel-trace summary of the
# bench command, HBM traffic counters (separate --pmc passes), the stage's per-tree kernels on the 8-section sample.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06i
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo rc=$?
tail -4 $O/pytest_gpu_full.txt
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; tail -c 400 $O/bench_c3.json
rocprofv3 --kernel-trace --stats -d $O/stats -o c3 -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --skip-chunk > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
python tools/rocprof_summary.py $(find $O/stats -name "*results.db" | head -1) > $O/kernel_stats_c3.txt 2>&1
rm -rf $O/stats
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > $O/bench_write.json 2> $O/write.err
python tools/pmc_summary.py $O > $O/pmc_c3.json 2> $O/pmc_summary.err
rm -rf $O/fetch $O/write
head -14 $O/kernel_stats_c3.txt; tail -c 1500 $O/pmc_c3.json
CHUNK_ROCPROF=$O/bt8 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/n5000_sample_under_rocprof.json 2> $O/bt8.err
python tools/rocprof_summary.py $(find $O/bt8 -name "*results.db" | head -1) > $O/kernel_stats_bounded_8_sections.txt 2>&1
rm -rf $O/bt8
head -14 $O/kernel_stats_bounded_8_sections.txt
