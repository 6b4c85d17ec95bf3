# Round 6, run E: the new full-length pins (C2, a C4 chunk, the C3 boundary windows) on the device; K2 of a whole
# window with and without the strip; the worker rule sweep at N = 2000 / 5000 / 10,000.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
timeout 1500 python -m pytest tests/test_full_pins_gpu.py tests/test_c3_full_gpu.py -x -q -m gpu > $O/pytest_pins.txt 2>&1; echo rc=$?
tail -5 $O/pytest_pins.txt
for ns in 1 2 0; do
  RELATE_AMD_REPAINT_NOSTRIP=$ns timeout 600 python bench.py --steps 2 --warmup 1 --skip-cpu --skip-alt --skip-chunk > $O/bench_k2_nostrip$ns.json 2> $O/bench_k2_nostrip$ns.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/bench_k2_nostrip$ns.json").read().strip().split("\n")[-1])
print("nostrip $ns", d.get("roofline_k2"), {k:v for k,v in d["config"].items() if "k2" in k.lower() or "repaint" in k.lower()})
PY
done
timeout 2400 python tools/worker_rule_sweep.py 2000 5000 10000 > $O/worker_rule.json 2> $O/worker_rule.err; echo rc=$?
cat $O/worker_rule.err | cut -c1-1500
