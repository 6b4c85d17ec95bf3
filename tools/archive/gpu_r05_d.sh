# Round 5, run D: the weave both ways + K3 with penalty and row minima: GPU tests of everything that builds trees, the
# per-tree kernels' times (rocprofv3 kernel stats of a bounded BuildTopology call), then the whole C3 chunk at 132 /
# 140 / 148 workers and at 140 with RePaint's descent as a kernel of its own.
export TMPDIR=/tmp
O=gpurun_out/r05d
mkdir -p $O
timeout 1500 python -m pytest tests/test_builder_gpu.py tests/test_builder_golden.py tests/test_builder_ages_gpu.py tests/test_stage_gpu.py tests/test_n5000_gpu.py tests/test_target_shard_gpu.py -x -q > $O/pytest_builders.txt 2>&1; echo rc=$?; tail -4 $O/pytest_builders.txt
CHUNK_ROCPROF=$O/stats RELATE_AMD_WINDOW_ROWS=32400 timeout 500 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/s8.json 2> $O/s8.err; echo rc=$?
python tools/rocprof_summary.py $(find $O/stats -name "*results.db" | head -1) > $O/kernel_stats_bounded_8_sections.txt 2>&1
rm -rf $O/stats
head -14 $O/kernel_stats_bounded_8_sections.txt
for w in 132 140 148; do
  RELATE_AMD_BUILD_WORKERS=$w timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w$w.json 2> $O/c3_w$w.err; echo rc=$?
done
RELATE_AMD_DESCENT_KERNEL=1 RELATE_AMD_BUILD_WORKERS=140 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w140_descent.json 2> $O/c3_w140_descent.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w132","c3_w140","c3_w148","c3_w140_descent"):
    try:
        d=json.load(open("gpurun_out/r05d/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:1], [l for l in d.get("stage_lines",[]) if l.startswith("[stage]")], d.get("section_md5",{}).get("out_133.anc"))
    except Exception as e: print(f, "failed", e)
PY
