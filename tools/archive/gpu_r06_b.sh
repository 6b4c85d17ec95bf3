# Round 6, run B: the re-laid tree worker (L_HOT: 13 B of LDS per cluster, two workgroups per CU): builder tests, then
# the 8-section N = 5000 sample with the phase timers, two per CU and one per CU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_builder_ages_gpu.py tests/test_n10000_gpu.py -x -q -m gpu > $O/pytest_builder.txt 2>&1; echo rc=$?
tail -5 $O/pytest_builder.txt
for occ in 2 1; do
  RELATE_AMD_BUILD_OCC=$occ RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/n5000_sample_occ$occ.json 2> $O/n5000_sample_occ$occ.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/n5000_sample_occ$occ.json").read().strip().split("\n")[-1])
print("occ $occ", d.get("build_topology_s"), d.get("gpu_builder_ms_per_tree"), d.get("md5",{}).get("out_0.anc"))
PY
done
