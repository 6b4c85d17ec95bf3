# Round 6, run R: all ten clusters' rows of phase A asked for at once (one memory round trip instead of two while more
# than 2560 clusters live) against five per pass: the 8-section sample, then the whole C3 chunk, alternating.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06r
mkdir -p $O
for i in 1 2; do
for V in qc5 qc10; do
  if [ $V = qc5 ]; then unset RELATE_EXE; else export RELATE_EXE=$PWD/relate_amd/variants/$V/Relate; fi
  RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/sample_${V}_$i.json 2> $O/sample_${V}_$i.err
  python - <<PY
import json
d=json.loads(open("$O/sample_${V}_$i.json").read().strip().split("\n")[-1])
t=d.get("gpu_builder_ms_per_tree") or {}
print("sample $V $i", round(sum(v for k,v in t.items() if k in ("updates","rescans","pair tests","pair order","ordered","symmetric","erase","pair scan")),1), t, d.get("md5",{}).get("out_0.anc"))
PY
done
done
for i in 1 2; do
for V in qc5 qc10; do
  if [ $V = qc5 ]; then unset RELATE_EXE; else export RELATE_EXE=$PWD/relate_amd/variants/$V/Relate; fi
  timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_${V}_$i.json 2> $O/c3_${V}_$i.err
  python - <<PY
import json
d=json.loads(open("$O/c3_${V}_$i.json").read().strip().split("\n")[-1])
print("C3 $V $i", round(d["wall_s"],1), d.get("section_md5",{}).get("out_133.anc"), (d.get("stage_summary") or [""])[0][-60:], d.get("gpu_builder_ms_per_tree"))
PY
done
done
