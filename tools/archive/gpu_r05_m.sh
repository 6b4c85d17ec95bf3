# Round 5, run M: the part launches' backward kernel without the LDS strip (RELATE_AMD_REPAINT_NOSTRIP=1: two waves
# a SIMD).  Same bits?  (every test that runs bounded windows, with the switch on.)  Then the whole C3 chunk.
export TMPDIR=/tmp
O=gpurun_out/r05m
mkdir -p $O
RELATE_AMD_REPAINT_NOSTRIP=1 timeout 1500 python -m pytest tests/test_window_gpu.py tests/test_n5000_gpu.py tests/test_c3_full_gpu.py tests/test_stage_gpu.py tests/test_target_shard_gpu.py -x -q > $O/pytest_nostrip.txt 2>&1; echo rc=$?; tail -3 $O/pytest_nostrip.txt
RELATE_AMD_REPAINT_NOSTRIP=1 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_nostrip_1.json 2> $O/e1.err; echo rc=$?
timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_1.json 2> $O/e2.err; echo rc=$?
RELATE_AMD_REPAINT_NOSTRIP=1 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_nostrip_1.json 2> $O/e3.err; echo rc=$?
RELATE_AMD_REPAINT_NOSTRIP=1 RELATE_AMD_BUILD_WORKERS=132 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w132_nostrip_1.json 2> $O/e4.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w116_nostrip_1","c3_w116_1","c3_w124_nostrip_1","c3_w132_nostrip_1"):
    try:
        d=json.load(open("gpurun_out/r05m/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
    except Exception as e: print(f, "failed", e)
PY
