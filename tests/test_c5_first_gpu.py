"""BASELINE.json config #5 (synthetic N = 10,000 x L = 200,000, seed 1, --memory 25: 327 windows) against the unmodified
reference at FULL length: tests/golden/c5_first.npz (tools/make_golden_full.py c5_first, hours of the reference in the
build container) holds the reference's complete paint file of window 0 -- md5 of the file and of every target's
record -- and `Relate --mode BuildTopology` of section 0 (md5 of .anc / .mut, every tree's parent array).

Here: FastPainting::PaintSteppingStones (fast_painting.cpp:18-618) at full length for a range of 192 targets on the
device (the two-wave tile of N > 5120; the stones of 192 targets are 5.5 GB) and their records of window 0 byte for
byte.  The section's trees are held to the fixture by the full-size job (tools/c5_job_one_gpu.py compares section 0's
md5; profiles/r06_c5_job_one_gpu.json) -- a Paint of all 10,000 targets is 2 x 131 GB of stones."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from relate_amd import api

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "c5_first.npz")


@pytest.fixture(scope="module")
def c5():
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/c5_first.npz not generated")
    z = np.load(GOLD)
    N, L, W, seed = [int(x) for x in z["meta"]]
    lib = api.lib()
    rw = (N + 31) // 32
    bits = np.zeros((L, rw), dtype=np.uint32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, None, bits.ctypes.data_as(C.c_void_p), rw, None,
                              r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = float(z["mem"][0]) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows_bits(N, L, bits.ctypes.data_as(C.c_void_p), rw, C.c_double(budget),
                                     wb.ctypes.data_as(C.c_void_p), 499) == W
    wb = wb[:W + 1].copy()
    assert np.array_equal(wb, z["wb"])
    return z, (N, L, W), bits, r, rpos, wb


@pytest.mark.parametrize("k0", [0, 5120, 9808])
def test_records_of_window_0_are_the_references_at_full_length(c5, k0):
    z, (N, L, W), bits, r, rpos, wb = c5
    ctx = api.Context()
    ctx.set_chunk_bits(N, bits, r, rpos, wb)
    ctx.set_target_range(k0, k0 + 192)
    ctx.prepare()
    ctx.paint(api.RL_SUM_EXACT)
    assert (ctx.N, ctx.L, ctx.W, ctx.waves) == (N, L, W, 2)
    bad = []
    for k in range(k0, k0 + 192):
        rec = ctx.paint_record(0, k)
        if len(rec) != int(z["s0/record_len"][k]) or hashlib.md5(rec).digest() != z["s0/record_md5"][k].tobytes():
            bad.append(k)
    ctx.close()
    assert not bad, "records of window 0 differ from the reference's for targets %s" % bad[:8]
