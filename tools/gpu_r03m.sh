# round 3, late: the C3 chunk with the CLI leaving its cache to the process exit, then the whole GPU suite and the smoke test
mkdir -p gpurun_out/r03m
C3_KEEP_STDERR=gpurun_out/r03m/c3.stderr timeout 420 python tools/chunk_c3_fused.py 267 > gpurun_out/r03m/c3.json 2> gpurun_out/r03m/c3.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03m/c3.json"))
print({k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines")})
PY
timeout 600 python -u -m pytest tests -x -q -m gpu > gpurun_out/r03m/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03m/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03m/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03m/smoke.txt
