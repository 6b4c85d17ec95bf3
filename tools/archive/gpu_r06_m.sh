# Round 6, run M: config #5's first window against the reference's records (tests/test_c5_first_gpu.py), and the
# CPU-side oracle on a few of them is not needed: the device is held to the reference directly.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06m
mkdir -p $O
timeout 1200 python -m pytest tests/test_c5_first_gpu.py -x -q -m gpu --durations=5 > $O/pytest_c5_first.txt 2>&1; echo rc=$?
tail -12 $O/pytest_c5_first.txt
