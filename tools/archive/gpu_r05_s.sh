# Round 5, run S (last): the whole C3 chunk with FindEquivalentBranches fused in (final library), bench --workload c4.
export TMPDIR=/tmp
O=gpurun_out/r05s
mkdir -p $O
C3_FUSED_FEB=1 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_fused_feb_final.json 2> $O/e1.err; echo rc=$?
timeout 300 python bench.py --workload c4 --skip-cpu > $O/bench_c4_one_gpu.json 2> $O/e2.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05s/c3_fused_feb_final.json"))
print("c3 + feb fused", round(d["wall_s"],1), d.get("stage_summary")[:1], d.get("fused_feb_lines"), d.get("section_md5"))
b=json.load(open("gpurun_out/r05s/bench_c4_one_gpu.json")); print(b["value"], b["ms_per_step"], b["config"]["workload"][:120])
PY
