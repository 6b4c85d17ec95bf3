# round 3, last experiment: fewer, larger windows now that the sections are dealt longest first (no even waves needed)
mkdir -p gpurun_out/r03q
RELATE_AMD_SECTION_THREADS=${ST:-112} RELATE_AMD_WINDOW_ROWS=${WR:-47000} C3_KEEP_STDERR=gpurun_out/r03q/c3.stderr timeout 420 python tools/chunk_c3_fused.py 267 > gpurun_out/r03q/c3_${ST:-112}.json 2> gpurun_out/r03q/c3.err; echo "rc=$?"
ST=${ST:-112} python - <<'PY'
import json,os
d=json.load(open("gpurun_out/r03q/c3_%s.json" % os.environ["ST"]))
print({k:d.get(k) for k in ("wall_s","trees_per_s","stage_lines")}); print(d["window_lines"][:2]); print(d["builder_host_side"][:1])
PY
