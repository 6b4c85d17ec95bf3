#!/usr/bin/env python3
"""Statistics of the serial normalising sums of PaintSteppingStones as the exact-sum kernels see them
(relate_amd/csrc/exact_sum.h): per lane of the 64-run layout, how often the true serial run crosses a binade,
meets a round-half-even tie, jumps two or more binades -- the cases that cost the kernels extra passes.
CPU only: the oracle paints a few targets with RO_SUM_PROBE and shows every term vector to this script.

    python tools/exact_sum_stats.py [N L targets]
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import rlutil  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 6

o = rlutil.oracle()
ch = rlutil.synth_chunk(N, L, seed=1, budget=None)
q, rem = N // 64, N % 64
lens = np.array([q + (1 if l < rem else 0) for l in range(64)])
starts = np.concatenate([[0], np.cumsum(lens)])[:64]
ends = starts + lens

stat = {d: dict(steps=0, cross=np.zeros(64), tie=np.zeros(64), both=np.zeros(64), jump=np.zeros(64),
                zero=np.zeros(64), rescale=0, crossf=np.zeros(64), tiepre=np.zeros(64)) for d in (0, 1)}


def expo(x):
    return np.frexp(x)[1]


def probe(ptr, n, d):
    t = np.ctypeslib.as_array(ptr, (n,)).copy()
    s = np.cumsum(t)  # numpy's cumsum of float64 is the serial left-to-right sum
    prev = np.concatenate([[0.0], s[:-1]])
    # exact rounding error of every addition (TwoSum)
    bb = s - prev
    err = (prev - (s - bb)) + (t - bb)
    e = expo(s)
    ulp = np.ldexp(1.0, e - 53)
    tie = (np.abs(err) == ulp / 2) & (t != 0)
    st = stat[d]
    st["steps"] += 1
    tot = s[-1]
    if tot < 1e-10 or tot > 1e10:
        st["rescale"] += 1
    e_in = expo(prev[starts])
    e_out = expo(s[ends - 1])
    zero_in = prev[starts] == 0.0
    cr = np.where(zero_in, 0, e_out - e_in)
    tl = np.add.reduceat(tie.astype(np.int64), starts) > 0
    tl &= ~zero_in
    st["zero"] += zero_in
    st["cross"] += cr >= 1
    st["jump"] += cr >= 2
    st["tie"] += tl
    st["both"] += (cr >= 1) & tl


PROBE = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_int)
cb = PROBE(probe)
o.ro_set_sum_probe(cb)
order = rlutil.RoSumOrder(2, 0, 0)
d = ch.ro()
W = ch.W
rng = np.random.RandomState(3)
for k in rng.randint(0, N, NT):
    bb = np.zeros(W, np.int32); be = np.zeros(W, np.int32)
    al = np.zeros((W, N), np.float32); bt = np.zeros((W, N), np.float32)
    la = np.zeros(W, np.float32); lb = np.zeros(W, np.float32)
    D = o.ro_paint_stepping_stones(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), W, int(k), C.byref(order),
                                   bb.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p),
                                   al.ctypes.data_as(C.c_void_p), bt.ctypes.data_as(C.c_void_p),
                                   la.ctypes.data_as(C.c_void_p), lb.ctypes.data_as(C.c_void_p))
    print("target", k, "D", D, flush=True)
np.set_printoptions(linewidth=200, precision=2, suppress=True)
for dname, dd in (("forward", 0), ("backward", 1)):
    st = stat[dd]
    n = st["steps"]
    print("==", dname, "steps", n, "rescales/step %.4f" % (st["rescale"] / n))
    for key in ("zero", "cross", "tie", "both", "jump"):
        v = st[key] / n
        print("  %-6s per step %.2f | lanes 0-7:" % (key, v.sum()), v[:8], "| 8-15 %.2f 16-31 %.2f 32-63 %.2f" %
              (v[8:16].sum(), v[16:32].sum(), v[32:].sum()))
