# Round 5, run R: the last library (forward kernel with the target loop, capped launches off): the tests that run
# RePaint in every form, and the whole C3 chunk once more.
export TMPDIR=/tmp
O=gpurun_out/r05r
mkdir -p $O
timeout 1200 python -m pytest tests/test_window_gpu.py tests/test_c3_full_gpu.py tests/test_n5000_gpu.py tests/test_golden_gpu.py tests/test_edge_gpu.py -x -q > $O/pytest_last.txt 2>&1; echo rc=$?; tail -3 $O/pytest_last.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_final.json 2> $O/e1.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05r/c3_final.json"))
print("c3_final", round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
PY
