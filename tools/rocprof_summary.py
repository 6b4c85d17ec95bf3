#!/usr/bin/env python3
"""Print the per-kernel summary (calls, total/average duration, VGPRs) of a
rocprofv3 results .db as text: `python tools/rocprof_summary.py x_results.db`"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary of", sys.argv[1])
print("%-70s %6s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-70s %6d %14.3f %14.3f %7.2f" % (name[:70], calls, total / 1e3, avg / 1e3, pct))
print()
print("%-70s %6s %6s %6s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch"))
for row in c.execute("select distinct name,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,grid_x,workgroup_x from kernels"):
    print("%-70s %6s %6s %6s %8s %8s  grid=%s wg=%s" % ((row[0][:70],) + tuple(row[1:])))
