# Round 6, run G: the whole C3 chunk with this round's library against round 5's (relate_amd/variants/r05, built from
# commit 3338474), alternating on one box; config #5's route (run_chunk_by_targets, one rank) at full length for 24
# sections.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06g
mkdir -p $O
for i in 1 2; do
  for lib in r06 r05; do
    if [ $lib = r05 ]; then export RELATE_EXE=$PWD/relate_amd/variants/r05/Relate; else unset RELATE_EXE; fi
    timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_${lib}_$i.json 2> $O/c3_${lib}_$i.err; echo rc=$?
    python - <<PY
import json
d=json.loads(open("$O/c3_${lib}_$i.json").read().strip().split("\n")[-1])
print("C3 $lib run $i", round(d["wall_s"],1), d.get("trees_built"), d.get("section_md5",{}).get("out_133.anc"), (d.get("stage_summary") or [""])[0][-170:], d.get("gpu_builder_ms_per_tree"))
PY
  done
done
unset RELATE_EXE
C5_SKIP_FUSED=1 timeout 1500 python tools/c5_job_one_gpu.py 24 24 > $O/c5_by_targets.json 2> $O/c5_by_targets.err; echo rc=$?
tail -c 1500 $O/c5_by_targets.json
