mkdir -p gpurun_out/r03e
START=$(date +%s); python3 bench.py > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err; echo "bench rc=$?"
echo "bench wall $(( $(date +%s) - START )) s"; tail -c 400 gpurun_out/r03e/bench.err | head -5
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/bench.json").read().strip().split("\n")[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","ms_per_step")})
print("roofline", d["roofline"]["frac"], d["config"]["kernel_ms"])
print("k2", d.get("roofline_k2",{}).get("frac"), d.get("roofline_k2",{}).get("ms"))
print("sample", {k:d["config"]["chunk_wallclock_sample"].get(k) for k in ("build_topology_s","trees","trees_per_s","error")})
print("full", d["config"].get("chunk_wallclock_c3"))
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="sample"})
PY
