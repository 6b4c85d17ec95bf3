# Round 5, run G: BASELINE.json config #4 as a job on one GPU (tools/c4_job_one_gpu.py): synthetic N = 2000 x 5M SNPs
# as .haps text -> MakeChunks --memory 1 -> every chunk through Paint + BuildTopology + FindEquivalentBranches (fused).
export TMPDIR=/tmp
O=gpurun_out/r05g
mkdir -p $O
df -h /tmp /dev/shm | tail -2 > $O/box.txt
timeout 2700 python tools/c4_job_one_gpu.py 2000 5000000 1 /tmp/c4job > $O/c4_job.json 2> $O/c4_job.err; echo rc=$?
tail -3 $O/c4_job.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05g/c4_job.json"))
print({k:v for k,v in d.items() if k not in ("per_chunk_s",)})
print(list(d.get("per_chunk_s",{}).items())[:4])
PY
