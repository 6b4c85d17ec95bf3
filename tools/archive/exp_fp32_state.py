#!/usr/bin/env python3
"""Experiment: would a Paint / RePaint kernel with FP32 per-donor state stay
inside the parity tolerance?  oracle/liboracle_fp32.so (make -C oracle liboracle_fp32.so) is the oracle with alpha /
beta rounded to float after every update -- sums, factors, logscales double, lane-run summation order -- run through
the whole path (stepping stones -> paint files -> RePaint -> GetMatrix) and compared with the REFERENCE's distance
matrices: the N = 70 fixture (whole matrices) and the N = 5000 chunk of tests/golden/n5000_matrix.npz (8 rows of 3
matrices).  Tolerance: |d - d_ref| <= 1e-5 * max(|d_ref|, max |logscale|).

    python tools/exp_fp32_state.py            (CPU only; N = 5000 takes a few minutes on 8 cores)
"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil
from golden_util import Fixture

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "liboracle_fp32.so"])


def load(name):
    lib = C.CDLL(os.path.join(ROOT, "oracle", name))
    lib.ro_paint_chunk.restype = C.c_int
    lib.ro_window_open.restype = C.c_void_p
    return lib


def matrices(lib, ch, snps_by_window, order_mode=4):
    """-> {(w, snp): matrix} from the library's own stones (lanes summation order)"""
    out = {}
    d = ch.ro()
    with tempfile.TemporaryDirectory() as tmp:
        assert lib.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, tmp.encode(), order_mode, 0, None,
                                  None) == 0
        M = np.zeros((ch.N, ch.N), np.float32)
        for w, snps in snps_by_window.items():
            s0 = int(ch.wb[w])
            win = lib.ro_window_open(C.byref(d), os.path.join(tmp, "relate_%d.bin" % w).encode(), s0, order_mode)
            cur = s0
            for s in snps:
                for t in range(cur + 1, s + 1):
                    lib.ro_window_advance(C.c_void_p(win), t)
                cur = s
                lib.ro_window_matrix(C.c_void_p(win), s, M.ctypes.data_as(C.c_void_p))
                out[(w, s)] = M.copy()
            lib.ro_window_free(C.c_void_p(win))
    return out


def report(tag, got, ref, scale, rows=None):
    worst, same, n = 0.0, 0, 0
    for key in ref:
        g = got[key] if rows is None else got[key][rows]
        r = ref[key]
        tol = 1e-5 * np.maximum(np.abs(r), scale)
        worst = max(worst, float((np.abs(g - r) / tol).max()))
        same += int((g.view(np.uint32) == r.view(np.uint32)).sum())
        n += r.size
    print("%-28s worst |d - d_ref| / tolerance %8.2f   identical entries %.4f" % (tag, worst, same / n))
    return worst


f64, f32 = load("liboracle.so"), load("liboracle_fp32.so")
with tempfile.TemporaryDirectory() as tmp:
    fx = Fixture("synth70", tmp)
    snps = {w: [s for s, _ in fx.matrices(w)] for w in fx.dump_windows()}
    ref = {(w, s): m for w in fx.dump_windows() for s, m in fx.matrices(w)}
    scale = max(1.0, max(float(np.abs(ls).max()) for w in fx.dump_windows() for ls, _ in fx.repaint(w)))
    print("N = 70 x L = 700 (synth70), max |logscale| %.1f:" % scale)
    report("  double state, lanes order", matrices(f64, fx.chunk, snps), ref, scale)
    report("  FLOAT state, lanes order", matrices(f32, fx.chunk, snps), ref, scale)

if "--small" not in sys.argv:
    zm = np.load(os.path.join(ROOT, "tests", "golden", "n5000_matrix.npz"))
    N, L, W, seed = [int(x) for x in zm["meta"]]
    from relate_amd import api
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8); bp = np.zeros(L, dtype=np.int32); r = np.zeros(L); rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = float(zm["mem"][0]) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499) == W
    ch = rlutil.Chunk(seq, r, rpos, wb[:W + 1].copy(), bp)
    rows = [int(x) for x in zm["rows"]]
    snps = {0: [int(x) for x in zm["snps"]]}
    ref = {(0, int(s)): zm["matrix_rows/%d" % i] for i, s in enumerate(zm["snps"])}
    scale = max(1.0, float(zm["logscale_max"][0]))
    print("N = 5000 x L = 1200 (n5000_matrix), window 0, max |logscale| %.1f:" % scale)
    report("  double state, lanes order", matrices(f64, ch, snps), ref, scale, rows)
    report("  FLOAT state, lanes order", matrices(f32, ch, snps), ref, scale, rows)
