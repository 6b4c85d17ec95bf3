# Round 5, run Q: 124 workers + part launches capped at 768 workgroups again (run P: 142.2 s), twice, and 128 workers.
export TMPDIR=/tmp
O=gpurun_out/r05q
mkdir -p $O
RELATE_AMD_REPAINT_GRID=768 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_g768_2.json 2> $O/e1.err; echo rc=$?
RELATE_AMD_REPAINT_GRID=768 RELATE_AMD_BUILD_WORKERS=128 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w128_g768.json 2> $O/e2.err; echo rc=$?
RELATE_AMD_REPAINT_GRID=768 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_g768_3.json 2> $O/e3.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w124_g768_2","c3_w128_g768","c3_w124_g768_3"):
    try:
        d=json.load(open("gpurun_out/r05q/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
    except Exception as e: print(f, "failed", e)
PY
