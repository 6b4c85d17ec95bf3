# Round 3, the late runs (one gpurun call each): STEP=c3 the whole C3 chunk through tools/chunk_c3_fused.py with the
# phase lines kept (settings through the environment: RELATE_AMD_BUILD_WORKERS, RELATE_AMD_SECTION_THREADS +
# RELATE_AMD_WINDOW_ROWS, RELATE_AMD_REPAINT_LANES; TAG names the output) | STEP=suite the whole GPU suite + smoke.
OUT=gpurun_out/r03late
mkdir -p $OUT
STEP=${STEP:-c3}
TAG=${TAG:-default}
if [ $STEP = c3 ]; then
  C3_KEEP_STDERR=$OUT/c3_$TAG.stderr timeout 420 python tools/chunk_c3_fused.py 267 > $OUT/c3_$TAG.json 2> $OUT/c3_$TAG.err; echo "rc=$?"
  TAG=$TAG OUT=$OUT python - <<'PY'
import json, os
d = json.load(open("%s/c3_%s.json" % (os.environ["OUT"], os.environ["TAG"])))
print({k: d.get(k) for k in ("wall_s", "trees_per_s", "stage_lines")}); print(d["window_lines"][:2]); print(d["builder_host_side"][:1])
PY
fi
if [ $STEP = suite ]; then
  timeout 700 python -u -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
fi
