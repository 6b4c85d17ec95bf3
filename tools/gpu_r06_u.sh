# Round 6, run U: smoke + builder / stage tests on the library as committed last.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06u
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1200 python -m pytest tests/test_builder_gpu.py tests/test_builder_ages_gpu.py tests/test_stage_gpu.py tests/test_env_switches.py tests/test_target_shard_gpu.py -x -q -m gpu > $O/pytest_last.txt 2>&1; echo rc=$?
tail -3 $O/pytest_last.txt
