# Round 6, run Q: the bench under the launcher with one rank (the driver's N > 1 command line at N = 1), the C4 workload
# line, and the target-sharded bench line with one rank.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06q
mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --skip-chunk --skip-cpu > $O/bench_under_launcher_1rank.json 2> $O/launcher.err; echo rc=$?
tail -c 600 $O/bench_under_launcher_1rank.json
timeout 600 python bench.py --workload c4 --skip-cpu > $O/bench_c4_one_gpu.json 2> $O/c4.err; echo rc=$?
tail -c 500 $O/bench_c4_one_gpu.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --shard targets --skip-chunk --skip-cpu > $O/bench_targets_1rank.json 2> $O/targets.err; echo rc=$?
tail -c 500 $O/bench_targets_1rank.json
