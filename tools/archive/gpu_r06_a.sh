# Round 6, run A: the box; SQ census of the Paint launch in all three modes at the C3 cut (N=5000 x L=100k) and at C2
# (N=1000 x L=100k) -> profiles/r06_sq_paint.txt; the fast modes' line of this round's library; one N=2000 chunk sample
# with the builder's phase timers (what a tree costs at C4's size).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06a
mkdir -p $O
(free -g; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /tmp . | tail -2; rocm-smi --showmeminfo vram | tail -4) > $O/box.txt 2>&1
for cfg in c3cut c2; do
  if [ $cfg = c3cut ]; then A="--snps 100000"; else A="--haplotypes 1000 --snps 100000 --memory 5"; fi
  mkdir -p $O/sq_$cfg
  for mode in exact lanes lanes32; do
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/sq_$cfg/$mode -o sq -- python3 bench.py --steps 1 --warmup 0 $A --mode $mode --skip-cpu --skip-alt --skip-k23 --skip-chunk > $O/sq_$cfg/${mode}_bench.json 2> $O/sq_$cfg/$mode.err
  done
done
python tools/sq_report.py $O/sq_c3cut "N=5000 x L=100k cut of C3 (tools/gpu_r06_a.sh)" > $O/sq_paint.txt 2>&1
python tools/sq_report.py $O/sq_c2 "C2, N=1000 x L=100k (tools/gpu_r06_a.sh)" >> $O/sq_paint.txt 2>&1
cat $O/sq_paint.txt
find $O -name "*.db" -delete
# the fast modes at C3 and C2 with this round's library (bench.py's other_modes carries lanes / lanes32)
timeout 900 python bench.py --steps 5 --warmup 1 --skip-cpu --skip-k23 --skip-chunk > $O/bench_c3_modes.json 2> $O/bench_c3_modes.err; echo rc=$?
timeout 300 python bench.py --steps 20 --warmup 2 --haplotypes 1000 --snps 100000 --memory 5 --skip-cpu --skip-k23 --skip-chunk > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$?
# a tree at N = 2000: 8 sections of an N=2000 x L=20000 chunk with the phase timers
RELATE_AMD_TIMING=1 timeout 300 python tools/chunk_wallclock_big.py 2000 20000 1 8 > $O/n2000_sample.json 2> $O/n2000_sample.err; echo rc=$?
grep -h "gpu tree builder" $O/n2000_sample.err | tail -3
tail -c 1500 $O/n2000_sample.json
cat $O/box.txt
