"""experiment: K tree builders on one GPU at once (one stream and one workgroup each): trees/s vs K"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
from test_builder_gpu import coalescent_matrix

N = int(sys.argv[1]); reps = int(sys.argv[2])
rng = np.random.RandomState(1)
d = coalescent_matrix(rng, N)
prior = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
for K in [int(x) for x in sys.argv[3:]]:
    bs = [api.Builder(N, device=0) for _ in range(K)]
    for b in bs: b.build(d, None)
    def work(b):
        for _ in range(reps): b.build(d, prior)
    th = [threading.Thread(target=work, args=(b,)) for b in bs]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "N", N, "K", K, "trees/s %.2f" % (K * reps / dt), "s/tree/builder %.3f" % (dt / reps), flush=True)
    for b in bs: b.close()
