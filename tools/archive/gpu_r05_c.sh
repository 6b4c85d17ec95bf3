# Round 5, run C: the C3 pin against the reference (tests/test_c3_full_gpu.py), then the whole C3 chunk at 116 / 124 /
# 132 workers again (run B's 124 was a slow run: which one is the setting, which the box?), the last one followed by
# FindEquivalentBranches with bulk, threaded file I/O (run B: 329 s with one fread per field).
export TMPDIR=/tmp
O=gpurun_out/r05c
mkdir -p $O
timeout 1200 python -m pytest tests/test_c3_full_gpu.py -x -q > $O/pytest_c3_full.txt 2>&1; echo rc=$?; tail -5 $O/pytest_c3_full.txt
timeout 900 python -m pytest tests/test_stage_gpu.py tests/test_pipeline_gpu.py -x -q > $O/pytest_stage.txt 2>&1; echo rc=$?; tail -3 $O/pytest_stage.txt
C3_FUSED_FEB=1 RELATE_AMD_BUILD_WORKERS=132 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_w132_fused_feb.json 2> $O/c3_w132_fused_feb.err; echo rc=$?
for w in 116 124; do
  RELATE_AMD_BUILD_WORKERS=$w timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w$w.json 2> $O/c3_w$w.err; echo rc=$?
done
C3_FEB=1 RELATE_AMD_BUILD_WORKERS=132 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_w132.json 2> $O/c3_w132.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w132_fused_feb","c3_w116","c3_w124","c3_w132"):
    try:
        d=json.load(open("gpurun_out/r05c/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:1], [l for l in d.get("stage_lines",[]) if l.startswith("[stage]")], d.get("find_equivalent_branches_s"), d.get("find_equivalent_branches_lines"), d.get("section_md5",{}).get("out_133.anc"), d.get("feb_md5"), d.get("fused_feb_lines"))
    except Exception as e: print(f, "failed", e)
PY
