"""The tree builder on the GPU (relate_amd/csrc/minmatch_gpu.hip) against the host builder -- which is the one
checked against the reference binary (tests/test_oracle_ref.py) and the golden tree sequences: the same parent
and child arrays, tree after tree, with the state MinMatch carries between builds."""
import numpy as np
import pytest

from relate_amd import api

pytestmark = pytest.mark.gpu


def tied_matrix(rng, N, ties=0.3):
    d = (rng.rand(N, N) * 4 + rng.rand(N)[:, None]).astype(np.float32)
    d[rng.rand(N, N) < ties] = 1.5  # plenty of exact ties: the random draws decide
    np.fill_diagonal(d, 0)
    return d


def coalescent_matrix(rng, N):
    """distances shaped like the path's: near-ultrametric from a random binary tree plus asymmetric noise"""
    order = rng.permutation(N)
    h = np.zeros((N, N), np.float32)
    groups = [[int(x)] for x in order]
    t = 0.0
    while len(groups) > 1:
        t += rng.exponential(1.0 / (len(groups) * (len(groups) - 1) / 2))
        a, b = sorted(rng.choice(len(groups), 2, replace=False))
        for x in groups[a]:
            for y in groups[b]:
                h[x, y] = h[y, x] = t
        groups[a] = groups[a] + groups[b]
        del groups[b]
    d = (h * 20 + rng.rand(N, N) * 0.05).astype(np.float32)
    d -= d.min(axis=1, keepdims=True)
    np.fill_diagonal(d, 0)
    return d


def run_sequence(N, mats, theta=0.001, all_on_gpu=True):
    host, dev = api.Builder(N, theta), api.Builder(N, theta, device=0)
    on_gpu = 0
    for t, (d, prior) in enumerate(mats):
        ref = host.build(d, prior)
        got = dev.build(d, prior)
        on_gpu += dev.last_on_gpu
        assert dev.last_on_gpu or not all_on_gpu  # (the symmetric fallback included)
        for name, a, b in zip(("parent", "child_left", "child_right"), ref, got):
            assert np.array_equal(a, b), (t, name, int(np.argmax(a != b)))
    host.close()
    dev.close()
    return on_gpu


@pytest.mark.parametrize("N,seed", [(5, 1), (64, 2), (130, 3), (260, 4), (1100, 5)])
def test_tied_matrices_with_and_without_prior(N, seed):
    rng = np.random.RandomState(seed)
    mats = [(tied_matrix(rng, N), None)]
    for t in range(3):  # row minima of the prior rise from tree to tree: the minima carried over stay below them
        mats.append((tied_matrix(rng, N), ((np.floor(rng.rand(N, N) * 4) + t) * 6.9).astype(np.float32)))
    mats.append((tied_matrix(rng, N, 0.0), None))
    # (a quarter of all pairs are candidates in these matrices: from N = 130 on a row has more mutually close
    #  partners than the pair scan keeps -- MM_HITS -- and such a tree is the host's)
    run_sequence(N, mats, all_on_gpu=N < 130)


@pytest.mark.parametrize("N,seed", [(90, 7), (400, 8), (1500, 9)])
def test_coalescent_matrices(N, seed):
    rng = np.random.RandomState(seed)
    mats = [(coalescent_matrix(rng, N), None)]
    for _ in range(2):
        prior = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
        mats.append((coalescent_matrix(rng, N), prior))
    assert run_sequence(N, mats) >= 1  # (not every tree fell back to the host)


def test_reference_unit_vectors_on_gpu():
    # include/test/test_treebuilder.cpp:9-139 (theta = 0.025)
    d5 = np.array([[0, 0, 1, 2, 2], [2, 0, 3, 4, 4], [0, 0, 0, 1, 1], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]], np.float32)
    b = api.Builder(5, 0.025, device=0)
    assert list(b.build(d5)[0][:8]) == [6, 6, 7, 5, 5, 8, 7, 8]
    b.close()
    d4 = np.array([[0, 1, 2, 2], [3, 0, 1, 1], [0, 1, 0, 1], [1, 1, 0, 0]], np.float32)
    b = api.Builder(4, 0.025, device=0)
    assert list(b.build(d4)[0][:6]) == [6, 5, 4, 4, 5, 6]
    b.close()


@pytest.mark.parametrize("N,seed", [(7, 1), (100, 2), (700, 3)])
def test_no_mutually_closest_pair_symmetric_fallback(N, seed):
    """circulant distances: every row's minimum points at a cluster whose own minimum points elsewhere, so the
    pairs come from the symmetric matrix (tree_builder.cpp:255-293, :968-1058) from the first merge on"""
    rng = np.random.RandomState(seed)
    idx = np.arange(N)
    circ = (((idx[None, :] - idx[:, None]) % N) * 10.0).astype(np.float32)
    mats = [(circ, None), (circ + rng.rand(N, N).astype(np.float32), None),
            (circ + np.floor(rng.rand(N, N) * 3).astype(np.float32), (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)),
            (tied_matrix(rng, N), None)]
    run_sequence(N, mats)


def test_too_many_tied_candidates_go_to_the_host():
    """a flat matrix: every pair is a candidate and every cluster rebuilds at the first merge -- more than the
    kernel's lists hold (status 2): the host builds that tree from the carried state, the next one is the device's"""
    N = 1300
    rng = np.random.RandomState(3)
    flat = np.full((N, N), 2.5, np.float32)
    np.fill_diagonal(flat, 0)
    mats = [(coalescent_matrix(rng, N), None), (flat, None), (coalescent_matrix(rng, N), flat * 2),
            (coalescent_matrix(rng, N), None)]
    assert run_sequence(N, mats, all_on_gpu=False) == 3


def split_tree_matrix(rng, N):
    """near-ultrametric distances from random recursive splits of a random leaf order (cheap at N = 5000),
    plus asymmetric noise -- the shape of the path's matrices"""
    order = rng.permutation(N)
    h = np.zeros((N, N), np.float32)
    stack = [(0, N, 1.0)]
    while stack:
        lo, hi, t = stack.pop()
        if hi - lo < 2:
            continue
        mid = lo + 1 + rng.randint(0, hi - lo - 1)
        a, b = order[lo:mid], order[mid:hi]
        h[np.ix_(a, b)] = t
        h[np.ix_(b, a)] = t
        stack.append((lo, mid, t * (0.55 + 0.4 * rng.rand())))
        stack.append((mid, hi, t * (0.55 + 0.4 * rng.rand())))
    d = (h * 20 + rng.rand(N, N) * 0.05).astype(np.float32)
    d -= d.min(axis=1, keepdims=True)
    np.fill_diagonal(d, 0)
    return d


@pytest.mark.parametrize("N", [5000, 5121, 5300])
def test_large_trees_state_in_lds_and_in_global_memory(N):
    """N = 5000: the headline size, per-cluster state in LDS; N > 5120: the state does not fit next to the pair
    lists, the same kernel on global arrays"""
    rng = np.random.RandomState(N)
    prior = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
    run_sequence(N, [(split_tree_matrix(rng, N), None), (split_tree_matrix(rng, N), prior)])


def test_builders_side_by_side_with_hand_overs_and_the_symmetric_pool(monkeypatch):
    """Several builders at once, as the sections of a stage: workers pull their trees from one queue.  One builder's
    flat matrix is handed to the host (status 2) WHILE the others' workgroups are building -- the status word
    must be out before the host hears of the tree, no end-of-launch write-back helps --, two builders
    need the symmetric matrix at the same time with ONE in the device's pool: one takes it, the other's tree goes to
    the host (status 1).  Every tree equals the host builder's."""
    import threading
    monkeypatch.setenv("RELATE_AMD_TEST_SYM_SLOTS", "1")
    N = 1300
    rng = np.random.RandomState(11)
    flat = np.full((N, N), 2.5, np.float32)
    np.fill_diagonal(flat, 0)
    idx = np.arange(N)
    circ = (((idx[None, :] - idx[:, None]) % N) * 10.0).astype(np.float32)
    plans = [[(coalescent_matrix(rng, N), None), (flat, None), (coalescent_matrix(rng, N), None)],
             [(circ, None), (circ + rng.rand(N, N).astype(np.float32), None)],
             [(circ, None), (coalescent_matrix(rng, N), None)],
             [(coalescent_matrix(rng, N), None), (coalescent_matrix(rng, N), flat * 2), (flat, None)]]
    want = []
    for mats in plans:
        host = api.Builder(N)
        want.append([host.build(d, p) for d, p in mats])
        host.close()
    devs = [api.Builder(N, device=0) for _ in plans]
    got = [None] * len(plans)
    on_gpu = [0] * len(plans)

    def work(b):
        out = []
        for d, p in plans[b]:
            out.append(devs[b].build(d, p))
            on_gpu[b] += devs[b].last_on_gpu
        got[b] = out

    threads = [threading.Thread(target=work, args=(b,)) for b in range(len(plans))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for b in range(len(plans)):
        assert got[b] is not None
        for t, (ref, g) in enumerate(zip(want[b], got[b])):
            for a, c in zip(ref, g):
                assert np.array_equal(a, c), (b, t)
    for d in devs:
        d.close()
    assert on_gpu[0] == 2 and on_gpu[3] == 2  # the flat matrices went to the host, the others stayed
    assert 2 <= on_gpu[1] + on_gpu[2] <= 4    # with one symmetric matrix in the pool a circulant tree may be the host's


@pytest.mark.parametrize("N,builders", [(300, 6), (2100, 3)])
def test_many_builders_side_by_side_build_the_same_trees(N, builders):
    """rl_debug_builder_throughput (the measurement hook of tools/bench_builder_many.py): several device builders, a host
    thread each, build one matrix (with a prior) three times side by side, the matrices staged from device memory --
    every builder's trees are builder 0's, and builder 0's first tree is the one a lone builder builds (N = 300: two
    workers per CU, N = 2100: one)"""
    import ctypes as C
    rng = np.random.RandomState(11)
    d = coalescent_matrix(rng, N)
    pr = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
    lib = api.lib()
    lib.rl_debug_builder_throughput.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
    secs, bad = C.c_double(0), C.c_int(-1)
    reps = 3
    first = np.zeros((reps, 2 * N - 1), np.int32)
    dd = np.ascontiguousarray(d, dtype=np.float32)
    rc = lib.rl_debug_builder_throughput(N, 0.001, 0, builders, reps, 0, dd.ctypes.data_as(C.c_void_p),
                                         pr.ctypes.data_as(C.c_void_p), C.byref(secs), C.byref(bad),
                                         first.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.rl_last_error().decode()
    assert bad.value == 0 and secs.value > 0.0
    b = api.Builder(N, device=0)
    for r in range(reps):  # (a builder carries its state from tree to tree: the same sequence)
        want = b.build(d.copy(), pr)[0]
        assert np.array_equal(first[r], want), "repetition %d" % r
    b.close()
