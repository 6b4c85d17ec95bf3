"""experiment: where do the stones of the HIP path and the oracle part ways"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rlutil
from relate_amd import api
from test_paint_gpu import oracle_stones
N, L, budget, seed = [int(x) for x in sys.argv[1:5]]
ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
for mode, name in ((api.RL_SUM_LANES, "lanes"), (api.RL_SUM_EXACT_SERIAL, "serial"), (api.RL_SUM_EXACT, "exact")):
    ctx = api.Context(); ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb); ctx.paint(mode)
    st = [ctx.stones(w) for w in range(ch.W)]
    for k in (0, 1, N - 1):
        bb, be, al, bt, la, lb = oracle_stones(ch, k, name == "lanes")
        for w in range(ch.W):
            da = np.flatnonzero(st[w]["alpha"][k].view(np.uint32) != al[w].view(np.uint32))
            db = np.flatnonzero(st[w]["beta"][k].view(np.uint32) != bt[w].view(np.uint32))
            print(name, "k", k, "w", w, "ls_a", st[w]["ls_alpha"][k], la[w], "ls_b", st[w]["ls_beta"][k], lb[w], "alpha diff", da[:6], "beta diff", db[:6])
            if len(db) and w == ch.W - 1:
                print("   beta gpu", st[w]["beta"][k][:8], "\n   beta ref", bt[w][:8])
            if len(da) and w <= 1:
                print("   alpha gpu", st[w]["alpha"][k][:8], "\n   alpha ref", al[w][:8])
    ctx.close()
