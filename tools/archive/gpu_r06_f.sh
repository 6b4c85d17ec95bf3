# Round 6, run F: SQ counters of the tree worker, one per CU and two per CU (one worker, one N = 5000 section); the
# chip's trees per second against the worker count (many builders side by side); the whole C3 chunk twice with the
# round's library; config #5 at full size as a job on one GPU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
for occ in 1 2; do
  CHUNK_PMC="$O/builder_pmc_occ$occ:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" RELATE_AMD_BUILD_OCC=$occ RELATE_AMD_GPU_BUILD=1 RELATE_AMD_BUILD_WORKERS=1 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 1 > $O/builder_pmc_occ$occ.json 2> $O/builder_pmc_occ$occ.err
  echo "== workers per CU: $occ" >> $O/builder_pmc.txt
  python tools/pmc_kernel.py $O/builder_pmc_occ$occ minmatch_worker >> $O/builder_pmc.txt 2>&1
  rm -rf $O/builder_pmc_occ$occ
done
cat $O/builder_pmc.txt
D=/tmp/mmdump; rm -rf $D; mkdir -p $D
RELATE_AMD_TEST_MM_DUMP=$D:3:5 RELATE_AMD_GPU_BUILD=1 timeout 300 python tools/chunk_wallclock_big.py 5000 20000 20 1 > /dev/null 2>&1
timeout 900 python tools/bench_builder_many.py $D 4 8:8:4 32:32:4 64:64:4 96:96:4 116:116:4 128:128:4 160:160:4 192:192:4 224:224:4 256:256:4 > $O/many_occ1.jsonl 2> $O/many_occ1.err; echo rc=$?
RELATE_AMD_BUILD_OCC=2 timeout 900 python tools/bench_builder_many.py $D 4 8:8:4 128:128:4 256:256:4 384:384:4 512:512:4 > $O/many_occ2.jsonl 2> $O/many_occ2.err; echo rc=$?
cat $O/many_occ1.jsonl $O/many_occ2.jsonl | cut -c1-260
for i in 1 2; do
  timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_run$i.json 2> $O/c3_run$i.err; echo rc=$?
  python - <<PY
import json
d=json.loads(open("$O/c3_run$i.json").read().strip().split("\n")[-1])
print("C3 run $i", round(d["wall_s"],1), d.get("trees_built"), d.get("section_md5"), d.get("stage_summary"), d.get("gpu_builder_ms_per_tree"))
PY
done
timeout 3300 python tools/c5_job_one_gpu.py 40 40 > $O/c5_job_one_gpu.json 2> $O/c5_job.err; echo rc=$?
tail -c 3000 $O/c5_job_one_gpu.json
tail -5 $O/c5_job.err | cut -c1-600
