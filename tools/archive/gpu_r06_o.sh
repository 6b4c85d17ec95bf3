# Round 6, run O: config #4 as a job on one GPU with the round's library (tools/c4_job_one_gpu.py: 20 GB of .haps text
# -> MakeChunks -> every chunk through Paint + BuildTopology + FindEquivalentBranches in one fused call).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06o
mkdir -p $O
timeout 2400 python tools/c4_job_one_gpu.py 2000 5000000 1 /tmp/c4job > $O/c4_job_one_gpu.json 2> $O/c4_job.err; echo rc=$?
tail -c 1500 $O/c4_job_one_gpu.json
tail -3 $O/c4_job.err | cut -c1-300
