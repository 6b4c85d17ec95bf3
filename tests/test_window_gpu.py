"""K2 (RePaintSection) and K3 (GetMatrix) parity: HIP vs the CPU oracle.

The oracle paints the chunk to paint files (byte-identical to the reference's,
see test_oracle_ref.py / test_oracle_golden.py); both sides then read the same
file, repaint the window and build distance matrices at several SNPs.
"""
import ctypes as C
import os

import numpy as np
import pytest

import rlutil
from relate_amd import api

pytestmark = pytest.mark.gpu


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run_case(tmp_path, N, L, budget, seed, theta=0.001, rho=1.0, windows=None, via_gpu_paint=False, chunk=None):
    o = rlutil.oracle()
    ch = chunk if chunk is not None else rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    if theta != 0.001 or rho != 1.0:
        ctx.set_painting(theta, rho)
        ch.theta = theta
        ch.r = ch.r * rho
    d = ch.ro()
    pdir = str(tmp_path)
    if via_gpu_paint:
        ctx.paint(api.RL_SUM_EXACT)
        ctx.write_paint_files(pdir)
    else:
        rc = o.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, pdir.encode(), 4, 0, None, None)
        assert rc == 0
    W = ch.W
    for w in (windows if windows is not None else sorted(set([0, W // 2, W - 1]))):
        pf = os.path.join(pdir, "relate_%d.bin" % w)
        s0, s1 = int(ch.wb[w]), int(ch.wb[w + 1]) - 1
        ow = o.ro_window_open(C.byref(d), pf.encode(), s0, 4) if o else None
        assert ow
        win = ctx.open_window(w, pf, s0, api.RL_SUM_EXACT)
        assert (win.start, win.end) == (o.ro_window_start(C.c_void_p(ow)), o.ro_window_end(C.c_void_p(ow)))
        for n in sorted(set([0, 1, N // 3, N - 1])):
            D = o.ro_window_rows(C.c_void_p(ow), n)
            assert win.rows(n) == D
            top, ls = win.topology(n)
            lso = np.ctypeslib.as_array(C.cast(o.ro_window_log(C.c_void_p(ow), n), C.POINTER(C.c_float)), (D,))
            topo = np.ctypeslib.as_array(C.cast(o.ro_window_top(C.c_void_p(ow), n), C.POINTER(C.c_float)), (D, N))
            assert np.array_equal(u32(ls), u32(lso)), (w, n, "logscales")
            assert np.array_equal(u32(top), u32(topo)), (w, n, "topology")
        # distance matrices at the window start and further in, cursors advanced
        # the way AncesTreeBuilder::BuildTopology does (anc_builder.cpp:487-495)
        M = np.zeros((N, N), np.float32)
        snps = sorted(set([s0, s0 + 1, s0 + (s1 - s0) // 3, s0 + 2 * (s1 - s0) // 3, s1]))
        cur = s0
        for s in snps:
            for t in range(cur + 1, s + 1):
                o.ro_window_advance(C.c_void_p(ow), t)
                win.advance(t)
            cur = s
            o.ro_window_matrix(C.c_void_p(ow), s, M.ctypes.data_as(C.c_void_p))
            G = win.matrix(s)
            assert np.array_equal(u32(G), u32(M)), (w, s, np.abs(G - M).max())
        win.close()
        o.ro_window_free(C.c_void_p(ow))
    ctx.close()


@pytest.mark.parametrize("N,L,budget,seed", [
    (8, 600, 3000, 3),
    (64, 1500, 30000, 1),
    (65, 1200, 30000, 2),
    (130, 1500, 200000, 5),
    (600, 900, 3000000, 11),
])
def test_window_matches_oracle(tmp_path, N, L, budget, seed):
    run_case(tmp_path, N, L, budget, seed)


def test_window_painting_params(tmp_path):
    run_case(tmp_path, 200, 1500, 400000, 7, theta=0.025, rho=3.0)


def test_gpu_paint_files_feed_window(tmp_path):
    # paint files written by the HIP Paint stage, read back by both sides
    run_case(tmp_path, 96, 1400, 60000, 9, via_gpu_paint=True)


@pytest.mark.parametrize("N", [5300, 7000, 9300])
def test_two_wavefronts_per_target(tmp_path, N):
    """N > 5120: RePaint and the matrix gather run on the 128-virtual-lane layout (two wavefronts per target); N = 9300
    is the S = 80 tile of that layout (72 KB of LDS strips per workgroup in the backward kernel)"""
    from test_edge_gpu import random_chunk
    ch = random_chunk(N, 70, 0.15, seed=5, wb=[0, 30, 70])
    run_case(tmp_path, N, 70, None, 5, chunk=ch, via_gpu_paint=True)


@pytest.mark.parametrize("N", [1000, 2000, 2100, 3500, 5000])
def test_single_wave_large_tiles(tmp_path, N):
    """K2 / K3 at the S = 48/64/80 register tiles (N = 5000: the headline configuration's): posterior rows,
    logscales and distance matrices of all three windows against the oracle, paint files by the oracle"""
    from test_edge_gpu import random_chunk
    ch = random_chunk(N, 330, 0.13, seed=N + 1, wb=[0, 100, 230, 330])
    run_case(tmp_path, N, 330, None, N, chunk=ch, windows=[0, 1, 2])


@pytest.mark.parametrize("N,L,wb", [(130, 400, [0, 150, 400]), (3500, 90, [0, 40, 90]), (5000, 90, [0, 40, 90]),
                                    (5300, 70, [0, 30, 70])])
def test_repaint_lanes_order_matches_oracle(tmp_path, N, L, wb):
    """RL_SUM_LANES in K2: posterior rows bit-identical to the oracle run in the same summation order
    (64 lane runs + balanced tree; 128 runs for the two-wavefront layout at N > 5120)"""
    from test_edge_gpu import random_chunk
    o = rlutil.oracle()
    ch = random_chunk(N, L, 0.15, seed=N, wb=wb)
    d = ch.ro()
    assert o.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, str(tmp_path).encode(), 4, 0, None,
                            None) == 0
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    order = rlutil.RoSumOrder(1, 0, 0)
    for w in range(ch.W):
        pf = os.path.join(str(tmp_path), "relate_%d.bin" % w)
        recs = rlutil.parse_paint_file(pf, N)
        win = ctx.open_window(w, pf, int(ch.wb[w]), api.RL_SUM_LANES)
        for n in sorted(set([0, 1, N // 2, N - 1])):
            r = recs[n]
            D = win.rows(n)
            top = np.zeros((D + 1, N), np.float32)
            ls = np.zeros(D + 1, np.float32)
            ab, be = np.ascontiguousarray(r["alpha"]), np.ascontiguousarray(r["beta"])
            assert o.ro_repaint_section(C.byref(d), ab.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p),
                                        r["bb"], r["be"], C.c_float(r["la"]), C.c_float(r["lb"]), n, C.byref(order),
                                        top.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p)) == D
            gtop, gls = win.topology(n)
            assert np.array_equal(u32(gls), u32(ls[:D])), (w, n)
            assert np.array_equal(u32(gtop), u32(top[:D])), (w, n)
        win.close()
    ctx.close()


@pytest.mark.parametrize("N,L,budget,seed,cap", [(96, 1400, 60000, 9, 0.08), (200, 1500, 400000, 7, 0.3),
                                                 (64, 1500, 30000, 1, 0.0), (5300, 500, 3.0e8, 3, 0.2)])
def test_bounded_window_same_matrices(tmp_path, N, L, budget, seed, cap):
    """rl_window_open_bounded: a window that keeps part of its posterior rows resident and repaints as the tree
    builder moves on gives, SNP by SNP, the matrices of the window that keeps everything (and of the oracle; the
    two-wavefront layout, N = 5300, against the full window only).  Later launches of a bounded window start their
    backward pass from the state an earlier one kept (repaint_kernels.hip)."""
    o = rlutil.oracle() if N <= 5120 else None
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    ctx.paint(api.RL_SUM_EXACT)
    ctx.write_paint_files(str(tmp_path))
    d = ch.ro()
    M = np.zeros((N, N), np.float32)
    for w in sorted(set([0, ch.W - 1])):
        pf = os.path.join(str(tmp_path), "relate_%d.bin" % w)
        s0, s1 = int(ch.wb[w]), int(ch.wb[w + 1]) - 1
        full = ctx.open_window(w, pf, s0, api.RL_SUM_EXACT)
        rows = sum(full.rows(n) for n in range(N))
        part = ctx.open_window(w, pf, s0, api.RL_SUM_EXACT, max_rows=max(1, int(cap * rows)))
        ow = o.ro_window_open(C.byref(d), pf.encode(), s0, 4) if o else None
        with pytest.raises(api.RelateError):
            part.topology(int(np.argmax([full.rows(n) for n in range(N)])))  # not all of its rows are there
        step = max(1, (s1 - s0) // 40)
        for s in range(s0, s1 + 1):
            if s > s0:
                full.advance(s)
                part.advance(s)
                if o:
                    o.ro_window_advance(C.c_void_p(ow), s)
            if (s - s0) % step == 0 or s == s1:
                A, B = full.matrix(s), part.matrix(s)
                assert np.array_equal(u32(A), u32(B)), (w, s)
                if o:
                    o.ro_window_matrix(C.c_void_p(ow), s, M.ctypes.data_as(C.c_void_p))
                    assert np.array_equal(u32(A), u32(M)), (w, s)
        assert full.repaints == 1
        assert part.repaints >= 2, part.repaints  # it did move through the window in parts
        part.close()
        full.close()
        if o:
            o.ro_window_free(C.c_void_p(ow))
    ctx.close()


def test_bounded_window_with_room_for_everything(tmp_path):
    ch = rlutil.synth_chunk(64, 900, seed=4, budget=30000)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    ctx.paint(api.RL_SUM_EXACT)
    win = ctx.open_window(0, None, int(ch.wb[0]), api.RL_SUM_EXACT, max_rows=10 ** 9)
    top, ls = win.topology(3)  # all rows resident: readable
    assert top.shape[0] == win.rows(3) and win.repaints == 1
    win.close()
    ctx.close()


def test_repaint_literal_serial_order_equals_the_parallel_exact_sums(tmp_path):
    """RePaint with RL_SUM_EXACT_SERIAL (the literal lane-after-lane order, the in-kernel fallback of the exact
    sums) gives the same posterior rows, logscales and matrices as RL_SUM_EXACT, bit for bit"""
    ch = rlutil.synth_chunk(130, 1500, seed=5, budget=200000)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    ctx.paint(api.RL_SUM_EXACT)
    for w in sorted(set([0, ch.W // 2, ch.W - 1])):
        s0 = int(ch.wb[w])
        a = ctx.open_window(w, None, s0, api.RL_SUM_EXACT)
        b = ctx.open_window(w, None, s0, api.RL_SUM_EXACT_SERIAL)
        for n in (0, 1, 64, 129):
            ta, la = a.topology(n)
            tb, lb = b.topology(n)
            assert np.array_equal(u32(ta), u32(tb)) and np.array_equal(u32(la), u32(lb)), (w, n)
        assert np.array_equal(u32(a.matrix(s0)), u32(b.matrix(s0)))
        a.close()
        b.close()
    ctx.close()
