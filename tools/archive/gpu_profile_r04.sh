# Round-4 evidence on the GPU box: smoke, the -m gpu suite, the default bench, rocprofv3 kernel stats of the bench,
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the Paint and RePaint kernels, SQ counters of the tree
# builder's workers on one section.  STEP=tests|bench|prof|builder picks a part (default: all).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04final
mkdir -p $OUT
STEP=${STEP:-all}
if [ $STEP = all ] || [ $STEP = tests ]; then
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
  cat $OUT/smoke.txt | tail -2; cat $OUT/pytest_gpu.txt
fi
if [ $STEP = all ] || [ $STEP = bench ]; then
  python3 bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_c3.json
fi
if [ $STEP = all ] || [ $STEP = prof ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o c3 -- python3 bench.py --steps 2 --warmup 1 --skip-cpu --skip-chunk > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
  python tools/rocprof_summary.py $(find $OUT/stats -name "*results.db" | head -1) > $OUT/kernel_stats_c3.txt 2>&1
  rm -rf $OUT/stats
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > $OUT/bench_fetch.json 2> $OUT/fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > $OUT/bench_write.json 2> $OUT/write.err
  python tools/pmc_summary_r03.py $OUT > $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/fetch $OUT/write
  head -12 $OUT/kernel_stats_c3.txt; tail -20 $OUT/pmc_summary.txt
fi
if [ $STEP = all ] || [ $STEP = builder ]; then
  # the workers of one N=5000 section (345 trees): SQ counters per dispatch of minmatch_worker
  CHUNK_PMC="$OUT/builder_pmc1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" RELATE_AMD_GPU_BUILD=1 RELATE_AMD_BUILD_WORKERS=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 1 > $OUT/builder_pmc1.json 2> $OUT/builder_pmc1.err
  python tools/pmc_kernel.py $OUT/builder_pmc1 minmatch_worker > $OUT/builder_pmc.txt 2>&1
  rm -rf $OUT/builder_pmc1
  cat $OUT/builder_pmc.txt | tail -14
fi
