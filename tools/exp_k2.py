"""experiment: K2 (RePaint) of one window with an alternative library build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
if sys.argv[1] != "default":
    api.LIB_PATH = os.path.join(ROOT, "build", sys.argv[1], "lib.so")
import bench
N, L, mem = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
mode = api.RL_SUM_LANES if len(sys.argv) > 5 and sys.argv[5] == "lanes" else api.RL_SUM_EXACT
bits, r, rpos, wb = bench.make_chunk(N, L, 1, mem)
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb)
ctx.paint(mode)
w = (len(wb) - 1) // 2
for rep in range(2):
    win = ctx.open_window(w, None, int(wb[w]), mode)
    print(sys.argv[1], "K2 ms", win.repaint_ms)
    win.close()
