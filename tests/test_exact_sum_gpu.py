"""RL_SUM_EXACT (relate_amd/csrc/exact_sum.h) must return the reference's
serial left-to-right double sum bit for bit -- on painting-like data and on
adversarial inputs: rounding ties, terms spanning many binades, jumps of
several binades, totals and prefixes next to powers of two, zeros."""
import ctypes as C

import numpy as np
import pytest

from relate_amd import api

pytestmark = pytest.mark.gpu


def gpu_sums(x, mode):
    x = np.ascontiguousarray(x, dtype=np.float64)
    batch, n = x.shape
    out = np.empty(batch, np.float64)
    rc = api.lib().rl_debug_wave_sum(x.ctypes.data_as(C.c_void_p), n, batch, mode, out.ctypes.data_as(C.c_void_p))
    assert rc == 0, api.lib().rl_last_error()
    return out


def serial_sums(x):
    return np.add.accumulate(np.asarray(x, np.float64), axis=1)[:, -1]  # sequential IEEE adds


def cases(n, batch, rng):
    u = rng.rand(batch, n)
    yield "uniform", u
    yield "lognormal wide", np.exp(rng.randn(batch, n) * 8.0)
    yield "painting-like", np.where(rng.rand(batch, n) < 0.02, rng.rand(batch, n), 1e-7 * rng.rand(batch, n))
    # many exact ties: small integers times a power of two against a big head
    t = rng.randint(0, 8, (batch, n)).astype(np.float64) * 2.0 ** -53
    t[:, 0] = 1.0
    yield "ties", t
    t2 = rng.randint(1, 4, (batch, n)).astype(np.float64) * 2.0 ** -52
    t2[:, 0] = 1.0 + 2.0 ** -52
    yield "ties odd head", t2
    j = 1e-12 * rng.rand(batch, n)
    for b in range(batch):
        j[b, rng.randint(0, n)] = 10.0 ** rng.randint(-3, 6)
        j[b, rng.randint(0, n)] = 10.0 ** rng.randint(-3, 6)
    yield "big jumps", j
    p = rng.rand(batch, n)
    p *= (2.0 ** rng.randint(-3, 4, (batch, 1))) / p.sum(axis=1, keepdims=True)  # totals ~ powers of two
    yield "total near power of two", p
    z = rng.rand(batch, n)
    z[:, : n // 3] = 0.0
    yield "leading zeros", z
    yield "all equal", np.full((batch, n), 0.1)
    yield "powers of two", 2.0 ** rng.randint(-30, 30, (batch, n)).astype(np.float64)


@pytest.mark.parametrize("n", [5, 63, 64, 200, 999, 3100, 4999, 5120])
def test_exact_sum_is_the_serial_sum(n):
    rng = np.random.RandomState(n)
    batch = 256 if n <= 1000 else 64
    for name, x in cases(n, batch, rng):
        ref = serial_sums(x)
        for mode, mname in ((api.RL_SUM_EXACT, "exact"), (api.RL_SUM_EXACT_SERIAL, "serial")):
            got = gpu_sums(x, mode)
            bad = np.nonzero(got.view(np.uint64) != ref.view(np.uint64))[0]
            assert len(bad) == 0, (n, name, mname, len(bad), got[bad[:3]], ref[bad[:3]])
        lanes = gpu_sums(x, api.RL_SUM_LANES)
        assert np.allclose(lanes, ref, rtol=1e-12, atol=0), (n, name, "lanes")
