# Round 6, run P: HBM bytes per tree of the tree worker (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) while
# 64 builders build one real N = 5000 tree's matrices side by side, 4 repetitions each (tools/bench_builder_many.py).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06p
mkdir -p $O
D=/tmp/mmdump; rm -rf $D; mkdir -p $D
RELATE_AMD_TEST_MM_DUMP=$D:3:5 RELATE_AMD_GPU_BUILD=1 timeout 300 python tools/chunk_wallclock_big.py 5000 20000 20 1 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/$c -o pmc -- python3 tools/bench_builder_many.py $D 4 64:64:4 > $O/many_$c.jsonl 2> $O/$c.err
  python - <<PY
import glob, sqlite3
db = sqlite3.connect(glob.glob("$O/$c/**/*.db", recursive=True)[0])
for k, n, tot in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name='$c' group by kernel_name"):
    print("$c", k[:60], "dispatches", n, "sum", tot)
PY
  cat $O/many_$c.jsonl
done
find $O -name "*.db" -delete
