#!/bin/bash
# experiment (round 3): per-tree phase times of the device builder against the number of trees in flight,
# with and without the scattered column store (RELATE_AMD_MM_DEBUG=2: wrong trees, timing only)
mkdir -p gpurun_out/conc
for dbg in 0 2; do
  for K in 1 32 64 96 128 192 256; do
    RELATE_AMD_MM_DEBUG=$dbg RELATE_AMD_TIMING=1 timeout 600 python3 tools/exp_builder_conc.py 5000 3 $K \
      > gpurun_out/conc/out_${dbg}_$K.txt 2> gpurun_out/conc/err_${dbg}_$K.txt
    python3 - <<PY
import re
acc={}; n=0
for l in open("gpurun_out/conc/err_${dbg}_$K.txt"):
    if "[gpu tree builder]" in l and "us:" in l:
        n+=1
        for m in re.finditer(r"([a-z_+ ]+?) (\d+)(?= |$)", l.split("us:")[1]):
            acc[m.group(1).strip()]=acc.get(m.group(1).strip(),0)+int(m.group(2))
tot=sum(v for k,v in acc.items() if k not in ("pairs_x100","shader_MHz"))
print("dbg",$dbg,"K",$K,"trees",n,"ms/tree %.1f"%(tot/max(n,1)/1000.0),{k:round(v/max(n,1)/1000.0,1) for k,v in acc.items()}, open("gpurun_out/conc/out_${dbg}_$K.txt").read().strip()[-60:])
PY
    rm -f gpurun_out/conc/err_${dbg}_$K.txt
  done
done
