"""Live check of the oracle against the reference binary on fresh random
chunks.  Needs oracle/_ref (built from /root/reference by `make -C oracle ref`),
so it only runs in the build container; the committed fixtures
(test_oracle_golden.py) carry the same evidence to the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

import rlutil

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not rlutil.have_ref(), reason="oracle/_ref not built (no /root/reference)")]


@pytest.mark.parametrize("N,L,budget,seed,painting", [
    (33, 1500, 20000, 31, None),
    (150, 2500, 500000, 32, None),
    (90, 1200, 100000, 33, "0.01,40"),   # large rho: r_prob clamp at 0.99
    (64, 4000, 60000, 34, "0.2,0.05"),   # large theta
])
def test_paint_files_match_reference_binary(tmp_path, oracle, N, L, budget, seed, painting):
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ch.write(str(tmp_path / "out"))
    args = ["--mode", "Paint", "--chunk_index", "0", "-o", "out"]
    if painting:
        args += ["--painting", painting]
        th, rho = painting.split(",")
        ch.theta = float(np.float32(th))
        ch.r = ch.r * float(np.float32(rho))
    rlutil.run_ref(args, cwd=str(tmp_path))
    d = ch.ro()
    out = tmp_path / "orc"
    out.mkdir()
    assert oracle.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, str(out).encode(), 4, 0, None,
                                 None) == 0
    for w in range(ch.W):
        a = open(tmp_path / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w), "rb").read()
        b = open(out / ("relate_%d.bin" % w), "rb").read()
        assert a == b, "window %d" % w


@pytest.mark.parametrize("N,L,budget,seed", [(160, 1800, 600000, 41), (300, 1200, 3000000, 42)])
def test_treeseq_matches_reference_binary(tmp_path, oracle, N, L, budget, seed):
    """host tree builder + tree-sequence loop (oracle as matrix provider) vs
    `Relate --mode BuildTopology` of the reference, byte for byte"""
    import test_treeseq_cpu as T

    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ch.write(str(tmp_path / "out"))
    rlutil.run_ref(["--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path))
    rlutil.run_ref(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                    str(ch.W - 1), "-o", "out"], cwd=str(tmp_path))

    class Fx:  # the minimal Fixture interface build_section needs
        pass
    fx = Fx()
    fx.chunk, fx.W, fx.dir = ch, ch.W, str(tmp_path / "out")
    fx.write_paint_files = lambda d: os.makedirs(d, exist_ok=True) or [
        os.replace(str(tmp_path / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w)),
                   os.path.join(d, "relate_%d.bin" % w))
        for w in range(ch.W) if not os.path.exists(os.path.join(d, "relate_%d.bin" % w))]
    for w in range(ch.W):
        anc, mut, nt = T.build_section(fx, w, tmp_path, oracle)
        assert mut == open(tmp_path / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read(), w
        assert anc == open(tmp_path / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read(), (w, nt)


@pytest.mark.parametrize("N,L,budget,seed", [(160, 1800, 600000, 41), (300, 1200, 3000000, 42)])
def test_find_equivalent_branches_matches_reference_binary(tmp_path, N, L, budget, seed):
    """the stage after BuildTopology (host code): same .anc files in, byte-identical .anc files out"""
    import shutil
    import subprocess

    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    ch.write(str(tmp_path / "out"))
    rlutil.run_ref(["--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path))
    rlutil.run_ref(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                    str(ch.W - 1), "-o", "out"], cwd=str(tmp_path))
    ours = tmp_path / "ours"
    ours.mkdir()
    shutil.copytree(str(tmp_path / "out"), str(ours / "out"))
    rlutil.run_ref(["--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path))
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "relate_amd", "Relate")
    p = subprocess.run([cli, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"], cwd=str(ours),
                       stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    for w in range(ch.W):
        a = open(ours / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read()
        b = open(tmp_path / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read()
        assert a == b, "window %d" % w


@pytest.mark.parametrize("seed,with_prior,threads", [(1, False, 1), (2, True, 1), (3, True, 4), (4, False, 4), (5, True, 4)])
def test_quickbuild_random_matrices_match_reference_binary(tmp_path, monkeypatch, seed, with_prior, threads):
    """MinMatch on unstructured matrices with many exact ties (and through the symmetric fallback), sequential and
    with helper threads, against MinMatch::QuickBuild of the reference"""
    import subprocess
    from relate_amd import api
    rng = np.random.RandomState(seed)
    N = 220
    d = (rng.rand(N, N) * 4 + rng.rand(N)[:, None]).astype(np.float32)
    d[rng.rand(N, N) < 0.3] = 1.5
    np.fill_diagonal(d, 0)
    d.tofile(str(tmp_path / "d.bin"))
    args = [rlutil.REF_HARNESS, "quickbuild", str(N), str(tmp_path / "d.bin"), str(tmp_path / "p.bin")]
    prior = None
    if with_prior:
        prior = (np.floor(rng.rand(N, N) * 4) * 6.9).astype(np.float32)
        prior.tofile(str(tmp_path / "prior.bin"))
        args.append(str(tmp_path / "prior.bin"))
    subprocess.run(args, check=True)
    ref = np.fromfile(str(tmp_path / "p.bin"), dtype=np.int32)
    monkeypatch.setenv("RELATE_AMD_BUILD_THREADS", str(threads))
    monkeypatch.setenv("RELATE_AMD_TEST_BUILD_MIN", "4")
    got = api.quickbuild(d.copy(), 0.001, prior)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("opts,flags,fb", [(["--no_consistency"], 1, 0), (["--fb", "3000"], 0, 3000),
                                           (["--fb", "1500", "--no_consistency"], 1, 1500)])
def test_treeseq_options_match_reference_binary(tmp_path, oracle, opts, flags, fb):
    """BuildTopology's --no_consistency (no carrier penalty / clade prior) and --fb (force a new tree every fb
    base pairs) against the reference binary, byte for byte"""
    import test_treeseq_cpu as T

    ch = rlutil.synth_chunk(90, 1500, seed=17, budget=200000)
    ch.write(str(tmp_path / "out"))
    rlutil.run_ref(["--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path))
    rlutil.run_ref(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                    str(ch.W - 1), "-o", "out"] + opts, cwd=str(tmp_path))

    class Fx:
        pass
    fx = Fx()
    fx.chunk, fx.W, fx.dir = ch, ch.W, str(tmp_path / "out")
    fx.write_paint_files = lambda d: os.makedirs(d, exist_ok=True) or [
        os.replace(str(tmp_path / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w)),
                   os.path.join(d, "relate_%d.bin" % w))
        for w in range(ch.W) if not os.path.exists(os.path.join(d, "relate_%d.bin" % w))]
    for w in range(ch.W):
        anc, mut, nt = T.build_section(fx, w, tmp_path, oracle, flags=flags, fb=fb)
        assert mut == open(tmp_path / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read(), w
        assert anc == open(tmp_path / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read(), (w, nt)


@pytest.mark.parametrize("seed,N", [(11, 60), (12, 220)])
def test_builder_sequences_match_one_reference_minmatch(tmp_path, seed, N):
    """rl_builder (host): a sequence of trees from ONE builder against a sequence from ONE MinMatch of the
    reference -- what MinMatch carries from build to build (min_values_CF, stale candidate indices) included.
    The GPU builder is held to the host builder on such sequences (tests/test_builder_gpu.py)."""
    import subprocess
    from relate_amd import api
    rng = np.random.RandomState(seed)
    mats = []
    for t in range(5):
        d = (rng.rand(N, N) * 4 + rng.rand(N)[:, None]).astype(np.float32)
        d[rng.rand(N, N) < 0.3] = 1.5
        np.fill_diagonal(d, 0)
        # priors whose row minima rise from tree to tree: the minima a builder carries over stay below them
        prior = None if t in (0, 3) else ((np.floor(rng.rand(N, N) * 4) + (t >= 2)) * 6.9).astype(np.float32)
        mats.append((d, prior))
    args = [rlutil.REF_HARNESS, "quickbuild_seq", str(N), str(tmp_path / "p.bin")]
    for t, (d, prior) in enumerate(mats):
        d.tofile(str(tmp_path / ("d%d.bin" % t)))
        args.append(str(tmp_path / ("d%d.bin" % t)))
        if prior is None:
            args.append("-")
        else:
            prior.tofile(str(tmp_path / ("c%d.bin" % t)))
            args.append(str(tmp_path / ("c%d.bin" % t)))
    subprocess.run(args, check=True)
    ref = np.fromfile(str(tmp_path / "p.bin"), dtype=np.int32).reshape(len(mats), 2 * N - 1)
    b = api.Builder(N)
    fresh_differs = 0
    for t, (d, prior) in enumerate(mats):
        got = b.build(d, prior)[0]
        assert np.array_equal(got, ref[t]), t
        fresh_differs += not np.array_equal(api.quickbuild(d, 0.001, prior), ref[t])
    b.close()
    assert fresh_differs > 0  # (the carried state matters: a fresh builder per tree gives other trees)


@pytest.mark.ref
@pytest.mark.skipif(not rlutil.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed,N,levels", [(21, 40, 1), (22, 90, 3), (23, 260, 5), (24, 2000, 4)])
def test_builder_with_sample_ages_matches_reference_minmatch(tmp_path, seed, N, levels):
    """MinMatch::QuickBuild with sample ages (--sample_ages: the third candidate key and the coalescence clock,
    tree_builder.cpp:7-22, 149-252, 601-965, 1123-1233, 1738-1841, 2073-2355, 2407-2531): sequences of trees from ONE
    builder, with and without prior, tied and coalescent-shaped matrices, against ONE MinMatch of the reference"""
    import subprocess
    from relate_amd import api
    from test_builder_gpu import coalescent_matrix, split_tree_matrix
    rng = np.random.RandomState(seed)
    # ancient samples: a few sampling times, most haplotypes modern
    ages = np.zeros(N)
    for lv in range(1, levels):
        ages[rng.rand(N) < 0.15] = lv * 400.0 * (1 + rng.randint(0, 3))
    mats = []
    for t in range(5):
        if t == 4:
            d = split_tree_matrix(rng, N) if N > 400 else coalescent_matrix(rng, N)
        else:
            d = (rng.rand(N, N) * 4 + rng.rand(N)[:, None]).astype(np.float32)
            d[rng.rand(N, N) < (0.3 if N < 1000 else 0.02)] = 1.5
            np.fill_diagonal(d, 0)
        prior = None if t in (0, 3) else ((np.floor(rng.rand(N, N) * 4) + (t >= 2)) * 6.9).astype(np.float32)
        mats.append((d, prior))
    ages.tofile(str(tmp_path / "ages.bin"))
    args = [rlutil.REF_HARNESS, "quickbuild_seq_ages", str(N), str(tmp_path / "p.bin"), str(tmp_path / "ages.bin")]
    for t, (d, prior) in enumerate(mats):
        d.tofile(str(tmp_path / ("d%d.bin" % t)))
        args.append(str(tmp_path / ("d%d.bin" % t)))
        if prior is None:
            args.append("-")
        else:
            prior.tofile(str(tmp_path / ("c%d.bin" % t)))
            args.append(str(tmp_path / ("c%d.bin" % t)))
    subprocess.run(args, check=True)
    ref = np.fromfile(str(tmp_path / "p.bin"), dtype=np.int32).reshape(len(mats), 2 * N - 1)
    b = api.Builder(N)
    b.set_sample_ages(ages)
    plain = api.Builder(N)
    differs = 0
    for t, (d, prior) in enumerate(mats):
        got = b.build(d, prior)[0]
        assert np.array_equal(got, ref[t]), (t, int(np.argmax(got != ref[t])))
        differs += not np.array_equal(plain.build(d, prior)[0], ref[t])
    b.close()
    plain.close()
    if levels > 1:
        assert differs > 0  # (the ages matter: the builder without them gives other trees)


@pytest.mark.parametrize("N,trees,moves", [(2000, 10, 5), (2000, 8, 200), (3000, 6, 1500)])
def test_find_equivalent_branches_on_random_tree_sequences(N, trees, moves):
    """the chain search of round 5 (a branch's candidates from the ancestors of ONE leaf of the other tree instead of
    whole size classes) against the reference binary where neighbouring trees differ a little, a lot, and almost
    everywhere: random binary trees, `moves` single-leaf prune-and-regraft moves between neighbours
    (tools/check_feb_synthetic_trees.py: both binaries on the same .anc files, byte for byte)"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_feb_synthetic_trees.py")
    p = subprocess.run([sys.executable, tool, str(N), str(trees), str(moves), "11"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-600:]
    assert "same bytes" in p.stdout.decode(), p.stdout.decode()
