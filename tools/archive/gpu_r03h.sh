mkdir -p gpurun_out/r03b
run() {
  tag=$1; shift
  env "$@" timeout 600 python tools/chunk_c3_fused.py 267 > gpurun_out/r03b/c3_$tag.json 2> gpurun_out/r03b/c3.err; echo "$tag rc=$?"
  TAG=$tag python - <<'PY'
import json,os
d=json.load(open("gpurun_out/r03b/c3_%s.json" % os.environ["TAG"]))
print(os.environ["TAG"], {k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines")})
print(d.get("window_lines")[:1]); print(d.get("builder_host_side")[:1])
PY
}
run w90 RELATE_AMD_BUILD_WORKERS=90
run w110 RELATE_AMD_BUILD_WORKERS=110
