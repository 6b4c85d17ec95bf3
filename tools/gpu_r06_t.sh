# Round 6, run T: the width of the woven matrix's column panels (64: a column's stores 1 KB apart; 8 / 16 / 32: 128 / 256 /
# 512 B apart, consecutive lines at 8) -- the 8-section N = 5000 sample per variant.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06t
mkdir -p $O
for V in 64 8 16 32 64 8; do
  if [ $V = 64 ]; then unset RELATE_EXE; else export RELATE_EXE=$PWD/relate_amd/variants/panel$V/Relate; fi
  RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/sample_$V.json 2> $O/sample_$V.err
  python - <<PY
import json
d=json.loads(open("$O/sample_$V.json").read().strip().split("\n")[-1])
t=d.get("gpu_builder_ms_per_tree") or {}
print("panel $V", round(d.get("build_topology_s",0),1), round(sum(v for k,v in t.items() if k in ("updates","rescans","pair tests","pair order","ordered","symmetric","erase","pair scan")),1), t, d.get("md5",{}).get("out_0.anc"))
PY
done
