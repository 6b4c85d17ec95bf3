"""Pins the oracle (oracle/relate_oracle.c) against outputs of the REAL
reference: byte-identical paint files, RePaintSection dumps and GetMatrix
dumps on the committed fixtures, plus the reference's own unit-test known
answers (include/test/test_painting.cpp, test_log.cpp)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import rlutil
from golden_util import Fixture

CASES = [("synth24", None), ("synth24_paint", (0.025, 2.0)), ("synth70", None), ("example8", None),
         ("synth40_noisy", None)]


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name,painting", CASES)
def test_paint_files_byte_identical(tmp_path, oracle, name, painting):
    fx = Fixture(name, tmp_path, painting)
    ch = fx.chunk
    d = ch.ro()
    out = tmp_path / "orc"
    out.mkdir()
    sites = C.c_longlong(0)
    rc = oracle.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, str(out).encode(), 2, 0, None,
                               C.byref(sites))
    assert rc == 0 and sites.value > 0
    for w in range(fx.W):
        assert open(out / ("relate_%d.bin" % w), "rb").read() == fx.paint_file(w), "window %d" % w


@pytest.mark.parametrize("name,painting", CASES)
def test_repaint_and_matrices_bit_identical(tmp_path, oracle, name, painting):
    fx = Fixture(name, tmp_path, painting)
    ch = fx.chunk
    d = ch.ro()
    pdir = str(tmp_path / "paint")
    fx.write_paint_files(pdir)
    N = fx.N
    for w in fx.dump_windows():
        s0 = int(ch.wb[w])
        win = oracle.ro_window_open(C.byref(d), os.path.join(pdir, "relate_%d.bin" % w).encode(), s0, 2)
        assert win
        for n, (ls, top) in enumerate(fx.repaint(w)):
            D = oracle.ro_window_rows(C.c_void_p(win), n)
            assert D == len(ls)
            lso = np.ctypeslib.as_array(C.cast(oracle.ro_window_log(C.c_void_p(win), n), C.POINTER(C.c_float)), (D,))
            topo = np.ctypeslib.as_array(C.cast(oracle.ro_window_top(C.c_void_p(win), n), C.POINTER(C.c_float)),
                                         (D, N))
            assert np.array_equal(u32(ls), u32(lso)) and np.array_equal(u32(top), u32(topo)), (w, n)
        M = np.zeros((N, N), np.float32)
        cur = s0
        for s, ref in fx.matrices(w):
            for t in range(cur + 1, s + 1):
                oracle.ro_window_advance(C.c_void_p(win), t)
            cur = s
            oracle.ro_window_matrix(C.c_void_p(win), s, M.ctypes.data_as(C.c_void_p))
            assert np.array_equal(u32(M), u32(ref)), (w, s)
        oracle.ro_window_free(C.c_void_p(win))


def test_fast_log_tolerance(oracle):
    # include/test/test_log.cpp:5-14
    x = 1e-4
    while x <= 1e4:
        assert abs(oracle.ro_fast_log(C.c_float(x)) - math.log(x)) < 0.007
        x *= 1.7


def test_reference_painting_known_answer(oracle):
    # include/test/test_painting.cpp:7-135: N=5, L=10, theta=0.025, r=0; the
    # posterior is constant along the sequence and encodes the integer
    # mismatch matrix d
    N, L, theta = 5, 10, 0.025
    cols = ["0110000000", "0110010100", "0100000000", "0000100000", "0000100000"]
    seq = np.array([[ord(cols[n][s]) for n in range(N)] for s in range(L)], dtype=np.uint8)
    dref = np.array([[0, 0, 1, 2, 2], [2, 0, 3, 4, 4], [0, 0, 0, 1, 1], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]])
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, theta)
    ntheta = 1.0 - theta
    prior_theta = np.float32(theta / (N - 1.0) - ntheta / (N - 1.0))
    prior_ntheta = np.float32(ntheta / (N - 1.0))
    rescale = oracle.ro_fast_log(C.c_float(theta / (1.0 - theta)))
    for k in range(N):
        ab = np.array([(1.0 if seq[0][k] > seq[0][n] else 0.0) * prior_theta + prior_ntheta for n in range(N)],
                      dtype=np.float32)
        be = np.ones(N, np.float32)
        top = np.zeros((L + 1, N), np.float32)
        ls = np.zeros(L + 1, np.float32)
        D = oracle.ro_repaint_section(C.byref(d), ab.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p), 0,
                                      L - 1, C.c_float(0), C.c_float(0), k, None,
                                      top.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p))
        normc = oracle.ro_fast_log(C.c_float(N - 1.0)) - D * oracle.ro_fast_log(C.c_float(ntheta))
        for l in range(D):
            assert abs(ls[0] - ls[l]) < 1e-5
            for n in range(N):
                assert abs(top[l][n] - top[0][n]) < 1e-5
                if n != k:
                    v = (oracle.ro_fast_log(C.c_float(top[l][n])) + ls[l] + normc) / rescale
                    assert dref[k][n] == round(v)


def test_stone_codec_roundtrip(oracle):
    rng = np.random.RandomState(3)
    N = 97
    v = np.repeat(rng.rand(20).astype(np.float32), rng.randint(1, 9, 20))[:N]
    v = np.resize(v, N).astype(np.float32)
    v[5] = 0.0
    v[6] = 0.0  # zeros never merge (min = 0)
    buf = (C.c_ubyte * oracle.ro_stone_max_bytes(N))()
    n = oracle.ro_encode_stone(v.ctypes.data_as(C.c_void_p), N, 17, C.c_float(-3.5), buf)
    out = np.zeros(N, np.float32)
    bs, ls = C.c_int(), C.c_float()
    m = oracle.ro_decode_stone(buf, C.c_size_t(n), N, out.ctypes.data_as(C.c_void_p), C.byref(bs), C.byref(ls))
    assert m == n and bs.value == 17 and ls.value == -3.5
    assert np.allclose(out, v, rtol=1.1e-3, atol=0) and out[5] == 0 and out[6] == 0
    # idempotent: encoding the decoded stone gives the same bytes
    buf2 = (C.c_ubyte * oracle.ro_stone_max_bytes(N))()
    n2 = oracle.ro_encode_stone(out.ctypes.data_as(C.c_void_p), N, 17, C.c_float(-3.5), buf2)
    assert bytes(buf[:n]) == bytes(buf2[:n2])
