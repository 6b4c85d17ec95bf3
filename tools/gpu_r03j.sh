# round 3, late: sanity of the worker kernel's last changes, then the C3 chunk with whole rounds of workers per XCD
mkdir -p gpurun_out/r03j
timeout 400 python -u -m pytest tests/test_builder_gpu.py tests/test_stage_gpu.py -x -q -m gpu 2>&1 | tail -3
run() {
  tag=$1; shift
  env "$@" timeout 420 python tools/chunk_c3_fused.py 267 > gpurun_out/r03j/c3_$tag.json 2> gpurun_out/r03j/c3_$tag.err; echo "$tag rc=$?"
  TAG=$tag python - <<'PY'
import json,os
d=json.load(open("gpurun_out/r03j/c3_%s.json" % os.environ["TAG"]))
print(os.environ["TAG"], {k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines")})
print(d.get("window_lines")[:1]); print(d.get("builder_host_side")[:1]); print(d.get("gpu_builder_ms_per_tree"))
PY
}
run w96 C3_X=1
run w112 RELATE_AMD_BUILD_WORKERS=112
