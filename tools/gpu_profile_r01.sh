set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
# kernel-trace stats of the SAME bench command at C2-size (short) and the default C3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/c3 -o c3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/prof/c3_bench.json 2> gpurun_out/prof/c3_bench.err
ls -R gpurun_out/prof | head -30
