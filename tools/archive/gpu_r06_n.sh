# Round 6, run N: the whole GPU suite on the round's last library.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06n
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu_full.txt 2>&1; echo rc=$?
tail -22 $O/pytest_gpu_full.txt
