"""experiment: one real distance matrix (synthetic panel, GPU path) for host-side MinMatch profiling"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from relate_amd import api
import bench
N, L = int(sys.argv[1]), int(sys.argv[2])
bits, r, rpos, wb = bench.make_chunk(N, L, 1, 20.0)
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb); ctx.paint()
w = (len(wb) - 1) // 2
win = ctx.open_window(w, None, int(wb[w]))
d = win.matrix(int(wb[w]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "matrix_%d.npy" % N), d.astype(np.float16) if N > 3500 else d)
print("saved", d.shape, float(d.min()), float(d.max()))
