"""Edge cases of the painting path (HIP vs oracle, bit for bit): tiny panels,
single-window and many-window chunks, targets with no / only derived sites,
monomorphic rows, windows that fall between two visited sites of a target."""
import ctypes as C

import numpy as np
import pytest

import rlutil
from relate_amd import api
from test_paint_gpu import bits_equal, oracle_stones

pytestmark = pytest.mark.gpu


def random_chunk(N, L, density, seed, wb=None, special=None):
    rng = np.random.RandomState(seed)
    seq = (rng.rand(L, N) < density).astype(np.uint8) + ord("0")
    if special == "flat_targets":
        seq[:, 0] = ord("0")            # target 0 never derived: visits SNP 0 and L-1 only
        seq[:, N - 1] = ord("1")        # last target derived everywhere
    if special == "mono_rows":
        seq[L // 3] = ord("0")
        seq[L // 2] = ord("1")
    bp = 1000 + np.cumsum(rng.randint(1, 200, L)).astype(np.int32)
    rpos = np.concatenate([bp, [bp[-1] + 100]]).astype(np.float64) * 1e-8
    r = np.maximum(np.diff(rpos), 1e-10) * 2500
    if wb is None:
        wb = [0, L]
    return rlutil.Chunk(seq, r, rpos, np.array(wb, np.int32), bp)


def check(ch, modes=("exact", "lanes")):
    for mode in modes:
        ctx = api.Context()
        ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
        ctx.paint({"exact": api.RL_SUM_EXACT, "lanes": api.RL_SUM_LANES, "serial": api.RL_SUM_EXACT_SERIAL}[mode])
        st = [ctx.stones(w) for w in range(ch.W)]
        for k in range(ch.N) if ch.N <= 12 else sorted(set([0, 1, ch.N // 2, ch.N - 2, ch.N - 1])):
            bb, be, al, bt, la, lb = oracle_stones(ch, k, mode == "lanes")
            for w in range(ch.W):
                assert st[w]["bsnp_begin"][k] == bb[w] and st[w]["bsnp_end"][k] == be[w], (mode, k, w)
                assert bits_equal(st[w]["ls_alpha"][k], la[w]) and bits_equal(st[w]["ls_beta"][k], lb[w]), (mode, k, w)
                assert bits_equal(st[w]["alpha"][k], al[w]) and bits_equal(st[w]["beta"][k], bt[w]), (mode, k, w)
        ctx.close()


@pytest.mark.parametrize("N", [5121, 7000, 10240])
def test_two_wavefronts_per_target(N):
    """N > 5120: a workgroup of two wavefronts paints one target (128 virtual lanes); the sums run over both"""
    ch = random_chunk(N, 90, 0.15, seed=N, wb=[0, 40, 90], special="flat_targets")
    check(ch, modes=("exact", "lanes", "serial"))


@pytest.mark.parametrize("N,S", [(1000, 16), (2000, 32), (2100, 48), (3500, 64), (5000, 80), (5120, 80)])
def test_single_wave_large_tiles(N, S):
    """N = 1000 (BASELINE.json config #2's N, S = 16), N = 2000 (config #4's N, S = 32) and N = 2049..5120: one wavefront per target with the S = 48/64/80 register tiles -- N = 5000 is the tile of
    the headline configuration (BASELINE.json config #3); K1 in all three sum orders against the oracle
    (fast_painting.cpp:18-618)"""
    ch = random_chunk(N, 420, 0.13, seed=N, wb=[0, 130, 300, 420], special="flat_targets")
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    assert (ctx.tile, ctx.waves) == (S, 1)
    ctx.close()
    check(ch, modes=("exact", "lanes", "serial"))


def test_two_wavefronts_long_run_with_rescales():
    ch = random_chunk(6000, 700, 0.12, seed=77, wb=[0, 150, 400, 700])
    check(ch)


@pytest.mark.parametrize("N", [2, 3, 5, 9])
def test_tiny_panels(N):
    check(random_chunk(N, 40, 0.3, N))


def test_minimal_length():
    check(random_chunk(6, 2, 0.5, 1))
    check(random_chunk(6, 3, 0.5, 2))


def test_flat_and_saturated_targets():
    check(random_chunk(20, 300, 0.2, 3, wb=[0, 50, 120, 121 + 11, 300], special="flat_targets"))


def test_monomorphic_rows_and_dense_windows():
    # windows of 12 SNPs: most windows fall between two visited sites of sparse targets
    L = 240
    check(random_chunk(33, L, 0.04, 4, wb=list(range(0, L, 12)) + [L], special="mono_rows"))


def test_dense_panel_many_rescales():
    # dense derived alleles + large theta: many mismatches -> frequent rescaling (1e-10 / 1e10)
    ch = random_chunk(70, 400, 0.5, 5, wb=[0, 100, 250, 400])
    ch.theta = 0.001
    check(ch)


def test_bad_arguments_are_reported():
    ctx = api.Context()
    ch = random_chunk(8, 30, 0.3, 7)
    with pytest.raises(api.RelateError):
        ctx.set_chunk(ch.seq, ch.r, ch.rpos, np.array([0, 10, 10, 30], np.int32))   # empty window
    with pytest.raises(api.RelateError, match="exceeds the largest"):
        big = random_chunk(10241, 4, 0.3, 7)                                         # N > 2 * 80 * 64
        ctx.set_chunk(big.seq, big.r, big.rpos, big.wb)
    with pytest.raises(api.RelateError):
        ctx.paint()                                                                  # nothing loaded
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    with pytest.raises(api.RelateError):
        ctx.set_painting(1.5, 1.0)                                                   # theta >= 1
    with pytest.raises(api.RelateError):
        ctx.stones(0)                                                                # not painted yet
    ctx.close()
