# Round 6, run S: the plane of d(a,b) beside the woven matrix (the rescans and first filters of a merge read it: a quarter
# of the line requests) -- builder tests, the 8-section sample, then the whole C3 chunk against the library before it
# (relate_amd/variants/prev), alternating.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06s
mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_builder_ages_gpu.py tests/test_n10000_gpu.py -x -q -m gpu > $O/pytest_builder.txt 2>&1; echo rc=$?
tail -3 $O/pytest_builder.txt
for V in new prev new prev; do
  if [ $V = new ]; then unset RELATE_EXE; else export RELATE_EXE=$PWD/relate_amd/variants/$V/Relate; fi
  RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/sample_$V.json 2> $O/sample_$V.err
  python - <<PY
import json
d=json.loads(open("$O/sample_$V.json").read().strip().split("\n")[-1])
t=d.get("gpu_builder_ms_per_tree") or {}
print("sample $V", round(d.get("build_topology_s",0),1), round(sum(v for k,v in t.items() if k in ("updates","rescans","pair tests","pair order","ordered","symmetric","erase","pair scan")),1), t, d.get("md5",{}).get("out_0.anc"))
PY
done
for i in 1 2; do
for V in new prev; do
  if [ $V = new ]; then unset RELATE_EXE; else export RELATE_EXE=$PWD/relate_amd/variants/$V/Relate; fi
  timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_${V}_$i.json 2> $O/c3_${V}_$i.err
  python - <<PY
import json
d=json.loads(open("$O/c3_${V}_$i.json").read().strip().split("\n")[-1])
print("C3 $V $i", round(d["wall_s"],1), d.get("section_md5",{}).get("out_133.anc"), (d.get("stage_summary") or [""])[0][-170:], d.get("gpu_builder_ms_per_tree"))
PY
done
done
