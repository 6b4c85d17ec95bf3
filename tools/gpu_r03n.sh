# round 3, late: two RePaint lanes -- the stage/window tests, then the C3 chunk
mkdir -p gpurun_out/r03n
timeout 400 python -u -m pytest tests/test_stage_gpu.py tests/test_window_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
C3_KEEP_STDERR=gpurun_out/r03n/c3.stderr timeout 420 python tools/chunk_c3_fused.py 267 > gpurun_out/r03n/c3.json 2> gpurun_out/r03n/c3.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03n/c3.json"))
print({k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines")}); print(d["window_lines"][:2]); print(d["builder_host_side"][:1])
PY
