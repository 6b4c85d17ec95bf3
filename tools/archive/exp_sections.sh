#!/bin/bash
# BuildTopology host-side experiment: section placement and helper threads (run on the GPU box)
mkdir -p gpurun_out
o=gpurun_out/exp_sections.txt
: > $o
cat /proc/loadavg >> $o
run() { echo "== $1" >> $o; shift; env "$@" timeout 900 python tools/chunk_wallclock_big.py 5000 20000 20 $SEC 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['build_topology_s'], d['trees']); [print(x[x.find('MinMatch'):][:150]) for x in d['build_topology_phases']]" >> $o; cat /proc/loadavg >> $o; }
SEC=1 run "pin0 s1" RELATE_AMD_PIN=0
SEC=1 run "pin1 s1" RELATE_AMD_PIN=1
SEC=8 run "pin1 s8 bt1" RELATE_AMD_PIN=1 RELATE_AMD_BUILD_THREADS=1
SEC=8 run "pin1 s8 bt4" RELATE_AMD_PIN=1 RELATE_AMD_BUILD_THREADS=4
