# Round 5, run A: the per-tree kernels after the fusion (rocprofv3 kernel stats of a bounded BuildTopology call, as
# tools/gpu_r04_bounded_stats.sh), then the whole C3 chunk at 104 (default) and 116 workers with the timing lines.
export TMPDIR=/tmp
O=gpurun_out/r05a
mkdir -p $O
df -h /tmp . | tail -3 > $O/box.txt; free -g | head -2 >> $O/box.txt; nproc >> $O/box.txt
CHUNK_ROCPROF=$O/stats RELATE_AMD_WINDOW_ROWS=32400 timeout 500 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/s8.json 2> $O/s8.err; echo rc=$?
python tools/rocprof_summary.py $(find $O/stats -name "*results.db" | head -1) > $O/kernel_stats_bounded_8_sections.txt 2>&1
rm -rf $O/stats
head -16 $O/kernel_stats_bounded_8_sections.txt
timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_default.json 2> $O/c3_default.err; echo rc=$?
RELATE_AMD_BUILD_WORKERS=116 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116.json 2> $O/c3_w116.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_default","c3_w116"):
    try:
        d=json.load(open("gpurun_out/r05a/%s.json"%f))
        print(f, d["wall_s"], d.get("trees_built"), d.get("gpu_builder_ms_per_tree"), d.get("stage_lines",[])[-3:], d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:2])
    except Exception as e: print(f, "failed", e)
PY
