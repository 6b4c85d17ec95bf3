"""Host tree-sequence logic (MinMatch + MapMutation + BuildTopology loop +
.anc/.mut writers) against the REAL reference's .anc/.mut on the golden
fixtures, byte for byte -- on the CPU: the distance-matrix provider plugged into
rl_treeseq_build is the oracle here (tests only; the product stage plugs the
GPU rl_window in, see test_stage_gpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import rlutil
from golden_util import Fixture
from relate_amd import api

MATRIX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float))
ADVANCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)


def build_section(fx, w, tmp_path, oracle, flags=0, fb=0):
    lib = api.lib()
    lib.rl_treeseq_create.restype = C.c_void_p
    ch = fx.chunk
    d = ch.ro()
    pdir = str(tmp_path / "refpaint")
    fx.write_paint_files(pdir)
    s0 = int(ch.wb[w])
    s1 = int(ch.wb[w + 1]) - 1 if w < fx.W - 1 else ch.L - 1
    win = oracle.ro_window_open(C.byref(d), os.path.join(pdir, "relate_%d.bin" % w).encode(), s0, 2)
    assert win

    def matrix(user, snp, out):
        oracle.ro_window_matrix(C.c_void_p(win), snp, out)
        return 0

    def advance(user, snp):
        oracle.ro_window_advance(C.c_void_p(win), snp)
        return 0

    bits = ch.bits()
    state = np.fromfile(os.path.join(fx.dir, "chunk_0.state"), dtype=np.int32, offset=4)
    ts = lib.rl_treeseq_create(ch.N, ch.L, bits.ctypes.data_as(C.c_void_p), bits.shape[1],
                               ch.rpos.ctypes.data_as(C.c_void_p), ch.bp.ctypes.data_as(C.c_void_p),
                               state.ctypes.data_as(C.c_void_p), C.c_double(ch.theta))
    assert ts
    mcb, acb = MATRIX_FN(matrix), ADVANCE_FN(advance)
    rc = lib.rl_treeseq_build(C.c_void_p(ts), s0, s1, mcb, acb, None, flags, fb)
    assert rc == 0, lib.rl_last_error()
    anc, mut = str(tmp_path / ("s%d.anc" % w)), str(tmp_path / ("s%d.mut" % w))
    assert lib.rl_treeseq_write(C.c_void_p(ts), anc.encode(), mut.encode()) == 0
    nt = lib.rl_treeseq_num_trees(C.c_void_p(ts))
    lib.rl_treeseq_destroy(C.c_void_p(ts))
    oracle.ro_window_free(C.c_void_p(win))
    return open(anc, "rb").read(), open(mut, "rb").read(), nt


@pytest.mark.parametrize("name", ["synth24", "synth70", "example8", "synth40_noisy"])
def test_anc_mut_byte_identical_to_reference(tmp_path, oracle, name):
    fx = Fixture(name, tmp_path)
    sections = range(fx.W) if fx.W <= 6 or name == "synth40_noisy" else sorted(set(list(range(0, fx.W, max(1, fx.W // 6))) + [fx.W - 1]))
    for w in sections:
        anc, mut, nt = build_section(fx, w, tmp_path, oracle)
        ref_anc, ref_mut = fx.z["anc/%d" % w].tobytes(), fx.z["mut/%d" % w].tobytes()
        assert mut == ref_mut, "section %d .mut differs" % w
        assert anc == ref_anc, "section %d .anc differs (%d trees)" % (w, nt)


def test_helper_threads_do_not_change_the_trees(tmp_path, oracle, monkeypatch):
    """a tree builder with helper threads (parallel distance updates inside a merge) writes the same bytes"""
    monkeypatch.setenv("RELATE_AMD_BUILD_THREADS", "3")
    monkeypatch.setenv("RELATE_AMD_TEST_BUILD_MIN", "4")
    fx = Fixture("synth70", tmp_path)
    for w in (0, fx.W // 2, fx.W - 1):
        anc, mut, nt = build_section(fx, w, tmp_path, oracle)
        assert mut == fx.z["mut/%d" % w].tobytes() and anc == fx.z["anc/%d" % w].tobytes(), w


@pytest.mark.parametrize("seed,with_prior", [(1, False), (2, True), (3, True)])
def test_quickbuild_same_tree_with_helper_threads(monkeypatch, seed, with_prior):
    """unstructured matrices: many clusters rebuild their candidates per merge (more than the 32 mask bits),
    frequent row-minimum re-scans -- the threaded split of a merge must still give the sequential result"""
    import numpy as np
    from relate_amd import api
    rng = np.random.RandomState(seed)
    N = 260
    d = (rng.rand(N, N) * 4 + rng.rand(N)[:, None]).astype(np.float32)
    d[rng.rand(N, N) < 0.3] = 1.5  # plenty of exact ties
    np.fill_diagonal(d, 0)
    prior = (np.floor(rng.rand(N, N) * 4) * 6.9).astype(np.float32) if with_prior else None
    monkeypatch.setenv("RELATE_AMD_BUILD_THREADS", "1")
    ref = api.quickbuild(d.copy(), 0.001, None if prior is None else prior.copy())
    monkeypatch.setenv("RELATE_AMD_BUILD_THREADS", "4")
    monkeypatch.setenv("RELATE_AMD_TEST_BUILD_MIN", "4")
    got = api.quickbuild(d.copy(), 0.001, None if prior is None else prior.copy())
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


def test_quickbuild_reference_unit_vectors():
    # include/test/test_treebuilder.cpp:9-139 (theta = 0.025)
    d5 = np.array([[0, 0, 1, 2, 2], [2, 0, 3, 4, 4], [0, 0, 0, 1, 1], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]], np.float32)
    assert list(api.quickbuild(d5, theta=0.025)[:8]) == [6, 6, 7, 5, 5, 8, 7, 8]
    d4 = np.array([[0, 1, 2, 2], [3, 0, 1, 1], [0, 1, 0, 1], [1, 1, 0, 0]], np.float32)
    assert list(api.quickbuild(d4, theta=0.025)[:6]) == [6, 5, 4, 4, 5, 6]
    # an all-zero matrix must still give a valid binary tree (test_treebuilder.cpp:22-29)
    p = api.quickbuild(np.zeros((5, 5), np.float32), theta=0.025)
    assert p[-1] == -1 and sorted(np.bincount(p[:-1])[5:]) == [2, 2, 2, 2]


@pytest.mark.parametrize("seed", [1, 2, 5489, 4294967295])
def test_device_builder_rng_restatement_matches_the_library(seed):
    """minmatch_gpu.hip restates std::mt19937 and libstdc++'s generate_canonical<double, 53> for the device; the
    same functions compiled for the host against the library, draw by draw (200 regenerations of the state)"""
    from relate_amd import api
    assert api.lib().rl_debug_rng_mismatches(seed, 125000) == 0


def test_stage_worker_rule():
    """treeseq.cpp: stage_worker_goal -- the counts the measured configurations ran with"""
    from relate_amd import api
    g = api.lib().rl_debug_stage_worker_goal
    assert g(256, 134, 1, 1) == 112   # C3: 134 open sections, bounded windows, one worker per CU: 29/64 of the CUs in whole XCD rounds
    assert g(256, 235, 1, 2) == 232   # a C4 chunk: 235 sections, bounded, two per CU
    assert g(256, 43, 1, 1) == 43     # C5 on one GPU: HBM admits 43 sections
    assert g(256, 53, 0, 1) == 53     # whole windows resident: a worker per section ...
    assert g(256, 400, 0, 1) == 224   # ... up to 7/8 of the CUs
    assert g(304, 400, 1, 1) == 136   # another chip: a share of ITS CUs
    assert g(256, 0, 1, 1) == 1
