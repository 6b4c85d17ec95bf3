"""experiment: event rates of sum_exact_fast (needs build/stats/librelate_stats.so, built with -DRL_STATS)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
api.LIB_PATH = os.path.join(ROOT, "build", "stats", "librelate_stats.so")
import bench
N, L = int(sys.argv[1]), int(sys.argv[2])
bits, r, rpos, wb = bench.make_chunk(N, L, 1, float(sys.argv[3]))
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb)
lib = api.lib(); print("loaded", lib._name, os.environ.get("RELATE_AMD_STATS"))
os.environ["RELATE_AMD_STATS"] = "1"
st = (C.c_ulonglong * 16)()
ms = ctx.paint(api.RL_SUM_EXACT)
ms = ctx.paint(api.RL_SUM_EXACT)
assert lib.rl_debug_stats(C.c_void_p(ctx._h), st) == 0
print("N", N, "L", L, "ms", ms, ctx.paint_times())
for name, o in (("forward", 0), ("backward", 8)):
    calls, fb, nonpure, special = st[o], st[o + 1], st[o + 2], st[o + 3]
    print("%s: sum calls %d  fallback steps %d (%.3g)  non-pure lanes/call %.3f  special lanes/call %.3f" %
          (name, calls, fb, fb / max(calls, 1), nonpure / max(calls, 1), special / max(calls, 1)))
    print("   cycles per sum: scan %.0f  four runs %.0f  classification+map scan %.0f  walk %.0f" % tuple(st[o + k] / max(calls, 1) for k in (4, 5, 6, 7)))
