# experiment: RePaint's checkpoint interval (rebuilds the library on the GPU box for each value, restores 4 at the end)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ck
for ck in 6 8; do
  make -C relate_amd/csrc clean > /dev/null 2>&1
  make -C relate_amd/csrc -j32 EXTRA=-DRL_REPAINT_CHECKPOINT=$ck > gpurun_out/ck/build_$ck.log 2>&1
  python -m pytest tests/test_window_gpu.py tests/test_golden_gpu.py -q -x -m gpu 2>&1 | tail -1
  rocprofv3 --kernel-trace --stats -d gpurun_out/ck/stats$ck -o c3 -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > gpurun_out/ck/b$ck.json 2>/dev/null
  echo "CK=$ck"; python tools/rocprof_summary.py $(find gpurun_out/ck/stats$ck -name "*results.db" | head -1) 2>&1 | grep "repaint_.*RepaintParams)  *2 "
  rm -rf gpurun_out/ck/stats$ck
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/ck/f$ck -o f -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/ck/w$ck -o w -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt > /dev/null 2>&1
  python tools/pmc_kernel.py gpurun_out/ck/f$ck repaint_; python tools/pmc_kernel.py gpurun_out/ck/w$ck repaint_
  rm -rf gpurun_out/ck/f$ck gpurun_out/ck/w$ck
done
