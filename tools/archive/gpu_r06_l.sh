# Round 6, run L: rows of rebuilt clusters per pass of a merge (2 / 3 / 4) now that a row costs 10 registers, not 20:
# the 8-section N = 5000 sample with the phase timers per variant (relate_amd/variants/rows<R>).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06l
mkdir -p $O
for R in 2 3 4 2 3 4; do
  RELATE_EXE=$PWD/relate_amd/variants/rows$R/Relate RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/rows$R.json 2> $O/rows$R.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/rows$R.json").read().strip().split("\n")[-1])
t=d.get("gpu_builder_ms_per_tree") or {}
print("rows $R", round(d.get("build_topology_s",0),1), round(sum(v for k,v in t.items() if k in ("updates","rescans","pair tests","pair order","ordered","symmetric","erase","pair scan")),1), t, d.get("md5",{}).get("out_0.anc"))
PY
done
