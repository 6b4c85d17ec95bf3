export TMPDIR=/tmp
mkdir -p gpurun_out/r03i
CHUNK_ROCPROF=gpurun_out/r03i/stats RELATE_AMD_WINDOW_ROWS=32400 timeout 500 python tools/chunk_wallclock_big.py 5000 20000 20 8 > gpurun_out/r03i/s8.json 2> gpurun_out/r03i/s8.err; echo rc=$?
python tools/rocprof_summary.py $(find gpurun_out/r03i/stats -name "*results.db" | head -1) > gpurun_out/r03i/kernel_stats_bounded_8_sections.txt 2>&1
rm -rf gpurun_out/r03i/stats
head -14 gpurun_out/r03i/kernel_stats_bounded_8_sections.txt
python -c "
import json; d=json.load(open('gpurun_out/r03i/s8.json')); print(d['build_topology_s'], d['trees'])"
