mkdir -p gpurun_out/r03b
true
[ -n "$PARTS" ] && export RELATE_AMD_WINDOW_PARTS=$PARTS
TAG=${TAG:-run}
timeout 700 python tools/chunk_c3_fused.py ${SECTIONS:-267} > gpurun_out/r03b/c3_$TAG.json 2> gpurun_out/r03b/c3.err; echo "c3 rc=$?"
tail -c 300 gpurun_out/r03b/c3.err; TAG=$TAG python - <<PY
import json,os
d=json.load(open("gpurun_out/r03b/c3_%s.json" % os.environ["TAG"]))
for k in ("wall_s","trees_built","trees_per_s","stage_lines","builder_host_side","gpu_builder_ms_per_tree","one_section","window_lines"):
    print(k, d.get(k) if k!="builder_host_side" else d.get(k)[:2])
PY
