# Round 5, run E: A/B on one box, alternating, 124 workers: (A) weave one way + separate penalty / row-minimum pass
# (the library of runs a-c) against (B) weave both ways + K3 with penalty and row minima (run d's, all ~171 s on its box).
export TMPDIR=/tmp
O=gpurun_out/r05e
mkdir -p $O
timeout 900 python -m pytest tests/test_stage_gpu.py tests/test_golden_gpu.py -x -q > $O/pytest_stage.txt 2>&1; echo rc=$?; tail -3 $O/pytest_stage.txt
for i in 1 2; do
  RELATE_AMD_WEAVE_ONE_WAY=1 RELATE_AMD_NO_K3_FUSION=1 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_A$i.json 2> $O/c3_w124_A$i.err; echo rc=$?
  RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_B$i.json 2> $O/c3_w124_B$i.err; echo rc=$?
done
python - <<'PY'
import json
for f in ("c3_w124_A1","c3_w124_B1","c3_w124_A2","c3_w124_B2"):
    try:
        d=json.load(open("gpurun_out/r05e/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:1], d.get("stage_summary"), d.get("sections_timeline",{}).get("sections_done_by_s"))
    except Exception as e: print(f, "failed", e)
PY
