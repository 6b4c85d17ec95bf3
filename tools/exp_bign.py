"""experiment: large-N single chunk (register tile S=128/160), parity on a few targets + timing"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rlutil
from relate_amd import api
from test_paint_gpu import oracle_stones, bits_equal
N, L, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ch = rlutil.synth_chunk(N, L, seed=3, budget=budget)
ctx = api.Context(); ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
print("N", N, "L", L, "W", ch.W, "sites", ctx.total_sites())
for mode, name in ((api.RL_SUM_EXACT, "exact"), (api.RL_SUM_LANES, "lanes")):
    ms = ctx.paint(mode)
    print(name, "kernel ms %.1f" % ms, "updates/s %.3g" % (2.0 * N * ctx.total_sites() / ms * 1e3))
    st = [ctx.stones(w) for w in range(ch.W)]
    for k in (0, N // 2, N - 1):
        bb, be, al, bt, la, lb = oracle_stones(ch, k, mode == api.RL_SUM_LANES)
        for w in range(ch.W):
            assert bits_equal(st[w]["alpha"][k], al[w]) and bits_equal(st[w]["beta"][k], bt[w]), (name, k, w)
            assert bits_equal(st[w]["ls_alpha"][k], la[w]) and bits_equal(st[w]["ls_beta"][k], lb[w]), (name, k, w)
    print(name, "parity ok")
