"""HIP path vs the committed golden fixtures (outputs of the REAL reference
binary): paint files byte-identical, RePaintSection rows and GetMatrix
matrices bit-identical, all through the C ABI."""
import os

import numpy as np
import pytest

from golden_util import Fixture
from relate_amd import api

pytestmark = pytest.mark.gpu

CASES = [("synth24", None), ("synth24_paint", (0.025, 2.0)), ("synth70", None), ("example8", None),
         ("synth40_noisy", None)]


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def open_ctx(fx):
    ctx = api.Context()
    ctx.load_chunk(fx.dir, 0)
    assert (ctx.N, ctx.L, ctx.W) == (fx.N, fx.L, fx.W)
    if fx.painting:
        th, rho = fx.painting
        ctx.set_painting(float(np.float32(th)), float(np.float32(rho)))
    return ctx


@pytest.mark.parametrize("name,painting", CASES)
def test_paint_files_byte_identical_to_reference(tmp_path, name, painting):
    fx = Fixture(name, tmp_path, painting)
    ctx = open_ctx(fx)
    ctx.paint(api.RL_SUM_EXACT)
    out = str(tmp_path / "paint")
    ctx.write_paint_files(out)
    for w in range(fx.W):
        assert open(os.path.join(out, "relate_%d.bin" % w), "rb").read() == fx.paint_file(w), "window %d" % w
    # one window's file alone, and the file as the sequence of its targets' records (rl_write_paint_file,
    # rl_paint_record: what PaintSteppingStones(data, wb, pfiles, k) appends to pfiles[w])
    w = fx.W // 2
    one = str(tmp_path / "one_window.bin")
    ctx.write_paint_file(w, one)
    assert open(one, "rb").read() == fx.paint_file(w)
    assert b"".join(ctx.paint_record(w, k) for k in range(ctx.N)) == fx.paint_file(w)
    ctx.close()


@pytest.mark.parametrize("name,painting", CASES)
def test_repaint_and_matrices_bit_identical_to_reference(tmp_path, name, painting):
    fx = Fixture(name, tmp_path, painting)
    ctx = open_ctx(fx)
    pdir = str(tmp_path / "refpaint")
    fx.write_paint_files(pdir)
    for w in fx.dump_windows():
        s0 = int(fx.chunk.wb[w])
        win = ctx.open_window(w, os.path.join(pdir, "relate_%d.bin" % w), s0, api.RL_SUM_EXACT)
        for n, (ls, top) in enumerate(fx.repaint(w)):
            assert win.rows(n) == len(ls)
            gtop, gls = win.topology(n)
            assert np.array_equal(u32(gls), u32(ls)) and np.array_equal(u32(gtop), u32(top)), (w, n)
        cur = s0
        for s, ref in fx.matrices(w):
            for t in range(cur + 1, s + 1):
                win.advance(t)
            cur = s
            assert np.array_equal(u32(win.matrix(s)), u32(ref)), (w, s)
        win.close()
    ctx.close()


@pytest.mark.parametrize("mode", [api.RL_SUM_LANES, api.RL_SUM_LANES32])
def test_lanes_mode_within_tolerance_of_reference(tmp_path, mode):
    # RL_SUM_LANES re-associates the normalising sums (RL_SUM_LANES32: and keeps the stepping-stone pass's state in
    # packed FP32): distances must stay
    # within |d_lanes - d_ref| <= 1e-5 * max(|d|, |logscale|) (SURVEY.md 7 H1:
    # the reference differs from itself by that much under FMA contraction)
    fx = Fixture("synth70", tmp_path)
    ctx = open_ctx(fx)
    ctx.paint(mode)
    for w in fx.dump_windows():
        s0 = int(fx.chunk.wb[w])
        win = ctx.open_window(w, None, s0, mode)
        scale = max(1.0, max(float(np.abs(ls).max()) for ls, _ in fx.repaint(w)))
        cur = s0
        for s, ref in fx.matrices(w):
            for t in range(cur + 1, s + 1):
                win.advance(t)
            cur = s
            g = win.matrix(s)
            tol = 1e-5 * np.maximum(np.abs(ref), scale)
            assert np.all(np.abs(g - ref) <= tol), (w, s, float(np.abs(g - ref).max()))
        win.close()
    ctx.close()
