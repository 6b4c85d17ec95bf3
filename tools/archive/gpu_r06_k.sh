# Round 6, run K: config #5's route with one rank / two ranks as threads again (worker expectations of the shards of one
# process now add up); a C4 chunk with two workers per CU against one.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06k
mkdir -p $O
timeout 900 python tools/chunk_c5_sharded.py 10000 20000 25 8 8 0.25 1 > $O/c5_route_one_rank.json 2> $O/c5_one.err; echo rc=$?
timeout 900 python tools/chunk_c5_sharded.py 10000 20000 25 8 4 0.25 2 > $O/c5_route_two_ranks.json 2> $O/c5_two.err; echo rc=$?
cat $O/c5_route_one_rank.json $O/c5_route_two_ranks.json
for i in 1 2; do
for occ in 2 1; do
  RELATE_AMD_BUILD_OCC=$occ timeout 600 python tools/chunk_c3_fused.py 9999 2000 121000 1 > $O/c4_chunk_occ${occ}_$i.json 2> $O/c4_occ${occ}_$i.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/c4_chunk_occ${occ}_$i.json").read().strip().split("\n")[-1])
print("C4 chunk occ $occ run $i", round(d["wall_s"],2), d.get("trees_built"), (d.get("stage_summary") or [""])[0][-100:])
PY
done
done
