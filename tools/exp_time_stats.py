"""experiment: per-segment cycle counters of the forward lanes-mode step (stats build with a patched kernel)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
api.LIB_PATH = os.path.join(ROOT, "build", "stats", "librelate_stats.so")
import bench
N, L = int(sys.argv[1]), int(sys.argv[2])
bits, r, rpos, wb = bench.make_chunk(N, L, 1, float(sys.argv[3]))
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb)
lib = api.lib()
os.environ["RELATE_AMD_STATS"] = "1"
st = (C.c_ulonglong * 16)()
ctx.paint(api.RL_SUM_LANES); ctx.paint(api.RL_SUM_LANES)
assert lib.rl_debug_stats(C.c_void_p(ctx._h), st) == 0
print("times", ctx.paint_times())
n = max(st[0], 1)
print("steps", st[0], "cycles/step: retire_touch %.0f | setup+slot %.0f | chunk loop %.0f | sum %.0f | tail %.0f" % tuple(st[k] / n for k in range(1, 6)))
