# Round 5, run P: strip-less part launches with at most RELATE_AMD_REPAINT_GRID workgroups (the kernels loop over the
# targets): does leaving CUs to the sections' own kernels let the stage carry more workers?
export TMPDIR=/tmp
O=gpurun_out/r05p
mkdir -p $O
RELATE_AMD_REPAINT_GRID=768 timeout 400 python -m pytest tests/test_window_gpu.py -x -q -k "bounded" > $O/pytest_grid.txt 2>&1; echo rc=$?; tail -2 $O/pytest_grid.txt
RELATE_AMD_REPAINT_GRID=768 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_g768.json 2> $O/e1.err; echo rc=$?
RELATE_AMD_REPAINT_GRID=1024 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_g1024.json 2> $O/e2.err; echo rc=$?
RELATE_AMD_REPAINT_GRID=896 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_g896.json 2> $O/e3.err; echo rc=$?
RELATE_AMD_REPAINT_GRID=640 RELATE_AMD_BUILD_WORKERS=132 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w132_g640.json 2> $O/e4.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w124_g768","c3_w124_g1024","c3_w116_g896","c3_w132_g640"):
    try:
        d=json.load(open("gpurun_out/r05p/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
    except Exception as e: print(f, "failed", e)
PY
