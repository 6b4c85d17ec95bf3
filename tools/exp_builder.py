"""experiment: tree builder on the GPU vs on the host -- how many trees stay on the GPU, time per tree"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
from test_builder_gpu import tied_matrix, coalescent_matrix

for kind, N, reps in [("tied", 260, 4), ("tied", 1100, 3), ("coal", 400, 3), ("coal", 1500, 3), ("coal", 5000, 2)]:
    rng = np.random.RandomState(N)
    host, dev = api.Builder(N), api.Builder(N, device=0)
    th = tg = 0.0
    on = 0
    same = True
    for t in range(reps):
        d = tied_matrix(rng, N) if kind == "tied" else coalescent_matrix(rng, N)
        prior = None if t == 0 else (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
        t0 = time.time(); a = host.build(d, prior); t1 = time.time(); b = dev.build(d, prior); t2 = time.time()
        th += t1 - t0; tg += t2 - t1; on += dev.last_on_gpu
        same = same and all(np.array_equal(x, y) for x, y in zip(a, b))
    print(kind, N, "trees", reps, "on gpu", on, "same", same, "host s/tree %.3f" % (th / reps), "gpu s/tree %.3f" % (tg / reps), flush=True)
