"""experiment: a few N=5000 trees on the GPU builder (for rocprofv3 --pmc runs of minmatch_kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from relate_amd import api
from test_builder_gpu import split_tree_matrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.RandomState(1)
mats = [split_tree_matrix(rng, N) for _ in range(2)]
prior = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
b = api.Builder(N, 0.001, device=0)
for t in range(T):
    t0 = time.time()
    b.build(mats[t % 2], prior if t else None)
    print("tree", t, "%.1f ms" % (1e3 * (time.time() - t0)), "on gpu", b.last_on_gpu, flush=True)
b.close()
