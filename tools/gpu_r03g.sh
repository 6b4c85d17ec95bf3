mkdir -p gpurun_out/r03b
RELATE_AMD_PARK_STONES=1 timeout 700 python tools/chunk_c3_fused.py 267 > gpurun_out/r03b/c3_parked.json 2> gpurun_out/r03b/c3.err; echo "c3 rc=$?"
tail -c 300 gpurun_out/r03b/c3.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03b/c3_parked.json"))
print({k:d.get(k) for k in ("wall_s","trees_built","trees_kept","trees_per_s","stage_lines")})
print(d.get("window_lines")[:2]); print(d.get("builder_host_side")[:1])
PY
