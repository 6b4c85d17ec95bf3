mkdir -p gpurun_out/r03b
C3_NO_TIMING=1 timeout 700 python tools/chunk_c3_fused.py 267 > gpurun_out/r03b/c3_notiming.json 2> gpurun_out/r03b/c3.err; echo "c3 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03b/c3_notiming.json"))
print({k:d.get(k) for k in ("wall_s","trees_built","trees_kept","trees_per_s")})
PY
timeout 700 python tools/chunk_c3_fused.py 267 > gpurun_out/r03b/c3_timing2.json 2> gpurun_out/r03b/c3.err; echo "c3 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03b/c3_timing2.json"))
print({k:d.get(k) for k in ("wall_s","trees_built","trees_kept","trees_per_s","stage_lines")})
PY
