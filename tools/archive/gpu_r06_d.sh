# Round 6, run D: the whole GPU suite on the re-laid worker; the 8-section N = 5000 sample (L_WARM, one per CU); a tree at
# N = 10,000 (L_HOT, 20 slots) and at N = 2000 (two per CU against one).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/n5000_sample.json 2> $O/n5000_sample.err; echo rc=$?
RELATE_AMD_TIMING=1 timeout 900 python tools/chunk_wallclock_big.py 10000 20000 25 4 > $O/n10000_sample.json 2> $O/n10000_sample.err; echo rc=$?
for occ in 2 1; do
RELATE_AMD_BUILD_OCC=$occ RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 2000 20000 1 8 > $O/n2000_sample_occ$occ.json 2> $O/n2000_sample_occ$occ.err; echo rc=$?
done
python - <<PY
import json
for f in ("n5000_sample","n10000_sample","n2000_sample_occ2","n2000_sample_occ1"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().split("\n")[-1])
        print(f, d.get("build_topology_s"), d.get("trees"), d.get("gpu_builder_ms_per_tree"), d.get("md5",{}).get("out_0.anc"))
    except Exception as e: print(f, "error", e)
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo rc=$?
tail -5 $O/pytest_gpu.txt
