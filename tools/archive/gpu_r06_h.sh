# Round 6, run H: builder tests on the final worker (DPP wave shift in the erase, batched state loads at 256 registers);
# the C3 A/B against round 5's library again; config #5's route with one rank, 24 sections, 1/20 of a window's rows kept.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_builder_ages_gpu.py tests/test_n10000_gpu.py tests/test_env_switches.py -x -q -m gpu > $O/pytest_builder.txt 2>&1; echo rc=$?
tail -4 $O/pytest_builder.txt
RELATE_AMD_TIMING=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/n5000_sample.json 2> $O/n5000_sample.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("$O/n5000_sample.json").read().strip().split("\n")[-1])
print("n5000 sample", d.get("build_topology_s"), d.get("gpu_builder_ms_per_tree"), d.get("md5",{}).get("out_0.anc"))
PY
for i in 1 2; do
  for lib in r05 r06; do
    if [ $lib = r05 ]; then export RELATE_EXE=$PWD/relate_amd/variants/r05/Relate; else unset RELATE_EXE; fi
    timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_${lib}_$i.json 2> $O/c3_${lib}_$i.err; echo rc=$?
    python - <<PY
import json
d=json.loads(open("$O/c3_${lib}_$i.json").read().strip().split("\n")[-1])
print("C3 $lib run $i", round(d["wall_s"],1), d.get("trees_built"), d.get("section_md5",{}).get("out_133.anc"), (d.get("stage_summary") or [""])[0][-120:], d.get("gpu_builder_ms_per_tree"))
PY
  done
done
unset RELATE_EXE
C5_SKIP_FUSED=1 timeout 1500 python tools/c5_job_one_gpu.py 24 24 > $O/c5_by_targets.json 2> $O/c5_by_targets.err; echo rc=$?
tail -c 1200 $O/c5_by_targets.json
