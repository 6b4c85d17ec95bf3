"""The oracle's PaintSteppingStones against the REFERENCE at the headline configuration's full length (N = 5000 x
L = 500,000, ~55,000 dependent steps per target, ~500 rescalings): the md5 of every (window, target) record the
reference's FastPainting::PaintSteppingStones wrote for a target (tests/golden/c3_full.npz, tools/make_golden_c3.py)
against the oracle's stones run through its encoder.  Four of the fixture's 16 targets here (the CPU suite's budget;
RELATE_C3_ORACLE_TARGETS=16 for all) -- the GPU suite holds the device to all 16 and to the oracle for 16 more."""
import ctypes as C
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import rlutil
from relate_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "c3_full.npz")


def test_oracle_records_at_full_length_are_the_references():
    import hashlib
    z = np.load(GOLD)
    N, L, W, seed = [int(x) for x in z["meta"]]
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    wb = np.ascontiguousarray(z["wb"], dtype=np.int32)
    o = rlutil.oracle()
    o.ro_encode_stone.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, 0.001)
    order = rlutil.RoSumOrder(0, 0, 0)
    cap = o.ro_stone_max_bytes(N)
    count = int(os.environ.get("RELATE_C3_ORACLE_TARGETS", "4"))
    picks = sorted(set(int(x) for x in np.linspace(0, len(z["targets"]) - 1, min(count, len(z["targets"])))))

    def one(ti):
        k = int(z["targets"][ti])
        a = np.zeros((W, N), np.float32)
        b = np.zeros((W, N), np.float32)
        la, lb = np.zeros(W, np.float32), np.zeros(W, np.float32)
        bb, be = np.zeros(W, np.int32), np.zeros(W, np.int32)
        assert o.ro_paint_stepping_stones(C.byref(d), wb.ctypes.data_as(C.c_void_p), W, k, C.byref(order),
                                          bb.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p),
                                          a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                          la.ctypes.data_as(C.c_void_p), lb.ctypes.data_as(C.c_void_p)) > 0
        buf = C.create_string_buffer(cap)
        bad = []
        for w in range(W):
            rec = struct.pack("<ii", int(wb[w]), int(wb[w + 1]) - 1)
            n = o.ro_encode_stone(a[w].ctypes.data_as(C.c_void_p), N, int(bb[w]), float(la[w]), buf)
            rec += buf.raw[:n]
            n = o.ro_encode_stone(b[w].ctypes.data_as(C.c_void_p), N, int(be[w]), float(lb[w]), buf)
            rec += buf.raw[:n]
            if hashlib.md5(rec).digest() != z["record_md5"][ti, w].tobytes():
                bad.append((k, w, float(la[w]), float(lb[w]), [float(x) for x in z["record_logscales"][ti, w]]))
        return bad

    with ThreadPoolExecutor(max_workers=min(len(picks), os.cpu_count() or 1)) as pool:
        bad = [x for res in pool.map(one, picks) for x in res]
    assert not bad, "%d records differ from the reference's; first: %s" % (len(bad), bad[:3])
