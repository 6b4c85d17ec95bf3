"""experiment: paint kernel times with an alternative library build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
if sys.argv[1] != "default":
    api.LIB_PATH = os.path.join(ROOT, "build", sys.argv[1], "lib.so")
import bench
N, L, mem = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
bits, r, rpos, wb = bench.make_chunk(N, L, 1, mem)
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb)
for mode, name in ((api.RL_SUM_EXACT, "exact"), (api.RL_SUM_LANES, "lanes")):
    ctx.paint(mode); ctx.paint(mode)
    print(sys.argv[1], name, "fwd %.1f ms bwd %.1f ms" % ctx.paint_times())
