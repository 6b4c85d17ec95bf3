"""RL_SUM_LANES32 -- the fast mode of K1: packed-FP32 state in the stepping-stone pass (paint32_kernels.hip).  Not
bit-identical to anything: its stepping stones are held to the FP64 `lanes` kernel's within a relative 2e-4 of each
stone's largest entry (100 - 1000 steps of float rounding; gross errors -- a wrong mask half, a lost slot, a missed
rescale -- are orders of magnitude above that) and its logscales within 1e-3; the distances against the REFERENCE are
in tests/test_golden_gpu.py (synth70) and tests/test_n5000_gpu.py (the headline tile), with the tolerance of
SURVEY.md 7 H1."""
import numpy as np
import pytest

import rlutil
from relate_amd import api
from test_edge_gpu import random_chunk

pytestmark = pytest.mark.gpu


def compare(ch):
    out = {}
    for name, mode in (("lanes", api.RL_SUM_LANES), ("lanes32", api.RL_SUM_LANES32)):
        ctx = api.Context()
        ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
        ctx.paint(mode)
        out[name] = [ctx.stones(w) for w in range(ch.W)]
        ctx.close()
    for w in range(ch.W):
        a, b = out["lanes"][w], out["lanes32"][w]
        assert np.array_equal(a["bsnp_begin"], b["bsnp_begin"]) and np.array_equal(a["bsnp_end"], b["bsnp_end"])
        for key in ("alpha", "beta"):
            scale = np.abs(a[key]).max(axis=1, keepdims=True)
            err = np.abs(a[key].astype(np.float64) - b[key]) / np.maximum(scale, 1e-300)
            assert np.isfinite(b[key]).all()
            assert err.max() <= 2e-4, (w, key, float(err.max()), np.unravel_index(err.argmax(), err.shape))
            assert np.array_equal(a[key] == 0, b[key] == 0) or np.abs(b[key][a[key] == 0]).max() <= 1e-30
        for key in ("ls_alpha", "ls_beta"):
            assert np.abs(a[key].astype(np.float64) - b[key]).max() <= 1e-3 * max(1.0, np.abs(a[key]).max() * 1e-3), (w, key)


@pytest.mark.parametrize("N,L,budget,seed", [(8, 600, 3000, 3), (64, 1500, 30000, 1), (65, 1200, 30000, 2),
                                             (130, 900, 200000, 4), (300, 2000, 400000, 5)])
def test_small_panels(N, L, budget, seed):
    compare(rlutil.synth_chunk(N, L, seed=seed, budget=budget))


@pytest.mark.parametrize("N", [1000, 2000, 2100, 3500, 5000, 5120])
def test_single_wave_tiles(N):
    compare(random_chunk(N, 420, 0.13, seed=N, wb=[0, 130, 300, 420], special="flat_targets"))


@pytest.mark.parametrize("N", [5121, 7000, 10240])
def test_two_wavefronts_per_target(N):
    compare(random_chunk(N, 90, 0.15, seed=N, wb=[0, 40, 90], special="flat_targets"))


def test_long_run_with_rescales():
    compare(random_chunk(700, 6000, 0.12, seed=77, wb=[0, 1500, 4000, 6000]))
