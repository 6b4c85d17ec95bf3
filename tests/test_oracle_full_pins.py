"""The oracle against the reference at FULL length for BASELINE.json config #2 (CPU): tests/golden/full_c2.npz holds the
md5 of every paint file the unmodified reference's `--mode Paint` wrote for the synthetic N = 1000 x L = 100,000 chunk
(tools/make_golden_full.py c2); the plain-C restatement (oracle/relate_oracle.c, fast_painting.cpp:18-618) paints the
same chunk on the host cores and must write the same bytes.  (The device path is held to the same fixture in
tests/test_full_pins_gpu.py.)"""
import ctypes as C
import hashlib
import os

import numpy as np

import rlutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_paints_config_2_as_the_reference_does(tmp_path):
    z = np.load(os.path.join(ROOT, "tests", "golden", "full_c2.npz"))
    N, L, W, seed = [int(x) for x in z["meta"]]
    budget = float(z["mem"][0]) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    from relate_amd import api
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499) == W
    wb = wb[:W + 1].copy()
    assert np.array_equal(wb, z["wb"])
    o = rlutil.oracle()
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, 0.001)
    out = str(tmp_path)
    assert o.ro_paint_chunk(C.byref(d), wb.ctypes.data_as(C.c_void_p), W, out.encode(), os.cpu_count() or 1, 0, None, None) == 0
    for w in range(W):
        b = open(os.path.join(out, "relate_%d.bin" % w), "rb").read()
        assert len(b) == int(z["paint_size"][w]), w
        assert hashlib.md5(b).digest() == z["paint_md5"][w].tobytes(), "paint file of window %d differs from the reference's" % w
