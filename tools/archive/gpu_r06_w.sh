# Round 6, run W: the launcher keeps whole XCD rounds only from 64 workers on (a stage of 43 sections had 40 workers,
# one of 53 had 48): builder / stage / env tests, the N = 5000 x L = 100k chunk (53 sections) again with the rule and
# its neighbours, and config #5 at full size again (43 sections open).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06w
mkdir -p $O
timeout 1200 python -m pytest tests/test_builder_gpu.py tests/test_stage_gpu.py tests/test_env_switches.py tests/test_n10000_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo rc=$?
tail -3 $O/pytest.txt
timeout 1200 python tools/worker_rule_sweep.py 5000 > $O/worker_rule_5000.json 2> $O/worker_rule.err; echo rc=$?
cat $O/worker_rule.err | cut -c1-900
C5_SKIP_ROUTE=1 timeout 2400 python tools/c5_job_one_gpu.py 0 0 > $O/c5_job_one_gpu.json 2> $O/c5_job.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("$O/c5_job_one_gpu.json").read().strip().split("\n")[-1])
f=d["fused_stage"]
print("C5", round(f["wall_s"],1), f.get("trees_built"), f.get("section_0_matches_reference"), f.get("stage_summary"))
PY
