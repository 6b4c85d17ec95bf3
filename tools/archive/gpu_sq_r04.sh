# Round 4: SQ issue / stall counters of the merged Paint launch in all three modes (exact, lanes, lanes32) on the
# L = 100k cut of C3 -> gpurun_out/r04sq/<mode>; tools/sq_report_r04.py prints them per forward + backward step pair.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04sq
for mode in exact lanes lanes32; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d gpurun_out/r04sq/$mode -o sq -- python3 bench.py --steps 1 --warmup 0 --snps 100000 --mode $mode --skip-cpu --skip-alt --skip-k23 --skip-chunk > gpurun_out/r04sq/${mode}_bench.json 2> gpurun_out/r04sq/$mode.err
done
python tools/sq_report_r04.py gpurun_out/r04sq > gpurun_out/r04sq/sq_paint.txt 2>&1
cat gpurun_out/r04sq/sq_paint.txt
find gpurun_out/r04sq -name "*.db" -delete
