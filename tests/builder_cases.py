"""Adversarial tree-builder sequences with the reference's answers on file (tests/golden/builder_adversarial.npz,
made by `python tools/make_golden.py builder` from the unmodified reference's MinMatch through oracle/ref_harness):
exact ties decided by the random draws, priors whose minima rise from tree to tree, the symmetric fallback from the
first merge on, a flat matrix (every pair a candidate: the device kernel hands that tree to the host), coalescent-
shaped matrices.  A case is a sequence of (distance matrix, prior or None) built by ONE builder, theta = 0.001."""
import numpy as np

from test_builder_gpu import coalescent_matrix, tied_matrix


def case_tied(N=64, seed=2):
    rng = np.random.RandomState(seed)
    mats = [(tied_matrix(rng, N), None)]
    for t in range(3):
        mats.append((tied_matrix(rng, N), ((np.floor(rng.rand(N, N) * 4) + t) * 6.9).astype(np.float32)))
    mats.append((tied_matrix(rng, N, 0.0), None))
    return N, mats


def case_symmetric(N=100, seed=2):
    rng = np.random.RandomState(seed)
    idx = np.arange(N)
    circ = (((idx[None, :] - idx[:, None]) % N) * 10.0).astype(np.float32)
    return N, [(circ, None), (circ + rng.rand(N, N).astype(np.float32), None),
               (circ + np.floor(rng.rand(N, N) * 3).astype(np.float32), (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)),
               (tied_matrix(rng, N), None)]


def case_flat(N=300, seed=3):
    rng = np.random.RandomState(seed)
    flat = np.full((N, N), 2.5, np.float32)
    np.fill_diagonal(flat, 0)
    return N, [(coalescent_matrix(rng, N), None), (flat, None), (coalescent_matrix(rng, N), flat * 2),
               (coalescent_matrix(rng, N), None)]


def case_coalescent(N=400, seed=8):
    rng = np.random.RandomState(seed)
    mats = [(coalescent_matrix(rng, N), None)]
    for _ in range(2):
        mats.append((coalescent_matrix(rng, N), (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)))
    return N, mats


CASES = {"tied": case_tied, "symmetric": case_symmetric, "flat": case_flat, "coalescent": case_coalescent}
