# Round 6, run X (last): the whole GPU suite and the default bench line on the library as committed last.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06x
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo rc=$?
tail -3 $O/pytest_gpu_full.txt
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; tail -c 300 $O/bench_c3.json
