# Round 4: kernel trace of a BuildTopology call with bounded windows (8 sections of the N = 5000 x L = 20k chunk, 1/31 of
# a window resident) -- the per-tree kernels and a RePaint launch on an otherwise idle chip (r03's record: prior_kernel
# 0.243 ms per tree).
export TMPDIR=/tmp
mkdir -p gpurun_out/r04ae
CHUNK_ROCPROF=gpurun_out/r04ae/stats RELATE_AMD_WINDOW_ROWS=32400 timeout 500 python tools/chunk_wallclock_big.py 5000 20000 20 8 > gpurun_out/r04ae/s8.json 2> gpurun_out/r04ae/s8.err; echo rc=$?
python tools/rocprof_summary.py $(find gpurun_out/r04ae/stats -name "*results.db" | head -1) > gpurun_out/r04ae/kernel_stats_bounded_8_sections.txt 2>&1
rm -rf gpurun_out/r04ae/stats
head -16 gpurun_out/r04ae/kernel_stats_bounded_8_sections.txt
