"""K1 parity: HIP PaintSteppingStones vs the CPU oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import rlutil
from relate_amd import api

pytestmark = pytest.mark.gpu


def oracle_stones(ch, k, lanes):
    o = rlutil.oracle()
    N, W = ch.N, ch.W
    bb = np.zeros(W, np.int32); be = np.zeros(W, np.int32)
    al = np.zeros((W, N), np.float32); bt = np.zeros((W, N), np.float32)
    la = np.zeros(W, np.float32); lb = np.zeros(W, np.float32)
    d = ch.ro()
    order = rlutil.RoSumOrder(1 if lanes else 0, 0, 0)
    D = o.ro_paint_stepping_stones(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), W, k, C.byref(order),
                                   bb.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p),
                                   al.ctypes.data_as(C.c_void_p), bt.ctypes.data_as(C.c_void_p),
                                   la.ctypes.data_as(C.c_void_p), lb.ctypes.data_as(C.c_void_p))
    assert D > 0
    return bb, be, al, bt, la, lb


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("N,L,budget,seed,theta,rho", [
    (8, 600, 3000, 3, 0.001, 1.0),
    (64, 1500, 30000, 1, 0.001, 1.0),
    (65, 1200, 30000, 2, 0.001, 1.0),
    (130, 1500, 200000, 5, 0.001, 1.0),
    (200, 2000, 400000, 7, 0.025, 3.0),     # exercises --painting
    (600, 1200, 2000000, 11, 0.001, 1.0),   # S=16 tile
    (1100, 700, 4000000, 13, 0.001, 50.0),  # S=32 tile, r_prob clamp
])
@pytest.mark.parametrize("mode", ["exact", "lanes", "serial"])
def test_paint_matches_oracle(N, L, budget, seed, theta, rho, mode):
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    if theta != 0.001 or rho != 1.0:
        ctx.set_painting(theta, rho)
        ch.theta = theta
        ch.r = ch.r * rho
    ctx.paint({"exact": api.RL_SUM_EXACT, "lanes": api.RL_SUM_LANES, "serial": api.RL_SUM_EXACT_SERIAL}[mode])
    W = ch.W
    st = [ctx.stones(w) for w in range(W)]
    targets = sorted(set([0, 1, N // 2, N - 1] + [int(x) for x in np.random.RandomState(seed).randint(0, N, 6)]))
    for k in targets:
        bb, be, al, bt, la, lb = oracle_stones(ch, k, mode == "lanes")
        for w in range(W):
            assert st[w]["bsnp_begin"][k] == bb[w] and st[w]["bsnp_end"][k] == be[w]
            assert bits_equal(st[w]["ls_alpha"][k], la[w]), (k, w, "ls_alpha")
            assert bits_equal(st[w]["ls_beta"][k], lb[w]), (k, w, "ls_beta")
            assert bits_equal(st[w]["alpha"][k], al[w]), (k, w, "alpha")
            assert bits_equal(st[w]["beta"][k], bt[w]), (k, w, "beta")
    ctx.close()
