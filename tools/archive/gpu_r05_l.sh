# Round 5, run L: RePaint's descent as a kernel of its own (two waves a SIMD) at 124 and 116 workers, alternating with
# the default on one box: does it move the edge?
export TMPDIR=/tmp
O=gpurun_out/r05l
mkdir -p $O
RELATE_AMD_DESCENT_KERNEL=1 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_descent_1.json 2> $O/e1.err; echo rc=$?
RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_1.json 2> $O/e2.err; echo rc=$?
RELATE_AMD_DESCENT_KERNEL=1 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_descent_1.json 2> $O/e3.err; echo rc=$?
timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_1.json 2> $O/e4.err; echo rc=$?
RELATE_AMD_DESCENT_KERNEL=1 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124_descent_2.json 2> $O/e5.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w124_descent_1","c3_w124_1","c3_w116_descent_1","c3_w116_1","c3_w124_descent_2"):
    try:
        d=json.load(open("gpurun_out/r05l/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("stage_cpu_s"), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
    except Exception as e: print(f, "failed", e)
PY
