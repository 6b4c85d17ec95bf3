# Round 6, run J: the batched exchange of run_chunk_by_targets -- its GPU tests, then config #5's N (L = 20,000, 8
# sections) through the route with one rank and with two ranks as threads of one process on the one GPU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06j
mkdir -p $O
timeout 900 python -m pytest tests/test_target_shard_gpu.py tests/test_n10000_gpu.py tests/test_c5_first_gpu.py -x -q -m gpu > $O/pytest_shard.txt 2>&1; echo rc=$?
tail -4 $O/pytest_shard.txt
timeout 900 python tools/chunk_c5_sharded.py 10000 20000 25 8 8 0.25 1 > $O/c5_route_one_rank.json 2> $O/c5_one.err; echo rc=$?
timeout 900 python tools/chunk_c5_sharded.py 10000 20000 25 8 4 0.25 2 > $O/c5_route_two_ranks.json 2> $O/c5_two.err; echo rc=$?
cat $O/c5_route_one_rank.json $O/c5_route_two_ranks.json
