# round 3, late: sections dealt longest first
mkdir -p gpurun_out/r03l
RELATE_AMD_BUILD_WORKERS=100 C3_KEEP_STDERR=gpurun_out/r03l/c3_w100.stderr timeout 420 python tools/chunk_c3_fused.py 267 > gpurun_out/r03l/c3_w100.json 2> gpurun_out/r03l/c3_w100.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03l/c3_w100.json"))
print({k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines","phases_s")})
PY
