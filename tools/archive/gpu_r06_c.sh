# Round 6, run C: the tree worker two per CU against one per CU -- the 8-section N = 5000 sample with the phase timers,
# then MANY builders side by side on one real tree's matrices (tools/bench_builder_many.py): trees per second of the
# whole chip with 256 workers (one per CU) and 512 (two per CU).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
for occ in 2 1; do
  RELATE_AMD_BUILD_OCC=$occ RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/n5000_sample_occ$occ.json 2> $O/n5000_sample_occ$occ.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/n5000_sample_occ$occ.json").read().strip().split("\n")[-1])
print("occ $occ", d.get("build_topology_s"), d.get("gpu_builder_ms_per_tree"), d.get("md5",{}).get("out_0.anc"))
PY
done
D=/tmp/mmdump; rm -rf $D; mkdir -p $D
RELATE_AMD_TEST_MM_DUMP=$D:3:5 RELATE_AMD_GPU_BUILD=1 timeout 300 python tools/chunk_wallclock_big.py 5000 20000 20 1 > /dev/null 2>&1
ls -la $D | head
RELATE_AMD_BUILD_OCC=1 timeout 900 python tools/bench_builder_many.py $D 4 8:8:4 128:128:4 256:256:4 > $O/many_occ1.jsonl 2> $O/many_occ1.err; echo rc=$?
cat $O/many_occ1.jsonl
timeout 900 python tools/bench_builder_many.py $D 4 8:8:4 128:128:4 256:256:4 384:384:4 512:512:4 > $O/many_occ2.jsonl 2> $O/many_occ2.err; echo rc=$?
cat $O/many_occ2.jsonl
tail -3 $O/many_occ2.err
