"""MinMatch::QuickBuild with sample ages (`--sample_ages`) on the GPU (minmatch_gpu.hip, the AGES build) against the
host builder with sample ages -- the one checked against the reference's MinMatch (tests/test_oracle_ref.py::
test_builder_with_sample_ages_matches_reference_minmatch) and its files (tests/golden/synth24_ages.npz): the same
parent and child arrays tree after tree from ONE builder, with the state it carries between builds."""
import os
import re
import subprocess

import numpy as np
import pytest

from relate_amd import api
from test_builder_gpu import coalescent_matrix, split_tree_matrix, tied_matrix

pytestmark = pytest.mark.gpu


def sample_ages(rng, N, levels):
    """ancient samples: a few sampling times, most haplotypes modern (levels = 0: every sample its own age -- no two
    candidates share the third key, and the clock reaches few of them)"""
    if levels == 0:
        return rng.permutation(N) * 37.0
    ages = np.zeros(N)
    for lv in range(1, levels):
        ages[rng.rand(N) < 0.15] = lv * 400.0 * (1 + rng.randint(0, 3))
    return ages


def run_sequence(N, ages, mats, theta=0.001, all_on_gpu=True):
    host, dev = api.Builder(N, theta), api.Builder(N, theta, device=0)
    host.set_sample_ages(ages)
    dev.set_sample_ages(ages)
    on_gpu = 0
    for t, (d, prior) in enumerate(mats):
        ref = host.build(d, prior)
        got = dev.build(d, prior)
        on_gpu += dev.last_on_gpu
        assert dev.last_on_gpu or not all_on_gpu
        for name, a, b in zip(("parent", "child_left", "child_right"), ref, got):
            assert np.array_equal(a, b), (t, name, int(np.argmax(a != b)))
    host.close()
    dev.close()
    return on_gpu


def priors(rng, N, t):
    return ((np.floor(rng.rand(N, N) * 4) + t) * 6.9).astype(np.float32)


@pytest.mark.parametrize("N,levels,seed", [(5, 2, 1), (40, 1, 2), (64, 3, 3), (90, 0, 4), (120, 5, 5), (260, 4, 6),
                                           (700, 0, 7), (1100, 3, 8)])
def test_tied_matrices_with_and_without_prior(N, levels, seed):
    """plenty of exact ties (the draws decide), priors whose minima rise from tree to tree (the minima carried over stay
    below them), trees with and without a prior in turn -- the clock starts differently for the two (:1155, :2440)"""
    rng = np.random.RandomState(seed)
    ages = sample_ages(rng, N, levels)
    mats = [(tied_matrix(rng, N), None)]
    for t in range(3):
        mats.append((tied_matrix(rng, N), priors(rng, N, t)))
    mats.append((tied_matrix(rng, N, 0.0), None))
    mats.append((tied_matrix(rng, N, 0.05), priors(rng, N, 1)))
    # (from N = 130 on a row of these matrices has more partners than the pair scan keeps: such a tree is the host's,
    #  and the device builder takes the state back for the next one)
    run_sequence(N, ages, mats, all_on_gpu=N < 130)


@pytest.mark.parametrize("N,levels,seed", [(90, 3, 11), (400, 0, 12), (400, 4, 13), (1500, 3, 14)])
def test_coalescent_matrices(N, levels, seed):
    rng = np.random.RandomState(seed)
    ages = sample_ages(rng, N, levels)
    mats = [(coalescent_matrix(rng, N), None)]
    for t in range(3):
        mats.append((coalescent_matrix(rng, N), (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)))
    mats.append((coalescent_matrix(rng, N), None))
    # (the bottom of these trees is one large near-tie: a merge in which every cluster rebuilds -- the first one of the
    #  leaf the last tree's candidates were renamed to, :2342 -- has more feasible pairs than the lists hold at N = 1500,
    #  and that tree is the host's)
    assert run_sequence(N, ages, mats, all_on_gpu=N < 1000) >= len(mats) - 2


@pytest.mark.parametrize("N,levels,seed", [(7, 2, 1), (100, 3, 2), (700, 4, 3)])
def test_no_mutually_closest_pair_symmetric_fallback(N, levels, seed):
    """circulant distances: the pairs come from the symmetric matrix (tree_builder.cpp:255-293, :968-1058), the
    candidates' clock runs beside it"""
    rng = np.random.RandomState(seed)
    ages = sample_ages(rng, N, levels)
    idx = np.arange(N)
    circ = (((idx[None, :] - idx[:, None]) % N) * 10.0).astype(np.float32)
    mats = [(circ, None), (circ + rng.rand(N, N).astype(np.float32), None),
            (circ + np.floor(rng.rand(N, N) * 3).astype(np.float32), (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)),
            (tied_matrix(rng, N), None)]
    run_sequence(N, ages, mats, all_on_gpu=N < 130)


@pytest.mark.parametrize("N", [5000, 5200])
def test_large_trees(N):
    """the headline size (ten register slots per thread) and one past it (twenty); the tree behind one with a prior
    begins with the candidates of the last one renamed up to its root (:2342): the first merge of that leaf sends
    every cluster through the rebuilding branch -- more than the list in LDS holds"""
    rng = np.random.RandomState(N)
    ages = sample_ages(rng, N, 4)
    prior = (np.floor(rng.rand(N, N) * 3) * 6.9).astype(np.float32)
    mats = [(split_tree_matrix(rng, N), None), (split_tree_matrix(rng, N), prior), (split_tree_matrix(rng, N), prior),
            (split_tree_matrix(rng, N), None)]
    # (these matrices are dense in feasible pairs -- 100 to 1000 per merge, 6 million in a tree --, and a merge of the
    #  N = 5200 sequence has more than the 32 N the lists hold: that tree is the host's)
    assert run_sequence(N, ages, mats, all_on_gpu=N == 5000) >= len(mats) - 1


def test_ages_set_again_and_all_equal():
    """other ages for the same builder (their table goes to the device again); all samples of one age: the clock is
    the only difference to a build without ages"""
    N = 150
    rng = np.random.RandomState(5)
    host, dev = api.Builder(N), api.Builder(N, device=0)
    for levels in (3, 1, 0):
        ages = sample_ages(rng, N, levels)
        host.set_sample_ages(ages)
        dev.set_sample_ages(ages)
        for t in range(3):
            d, prior = coalescent_matrix(rng, N), (None if t == 0 else priors(rng, N, t))
            assert np.array_equal(host.build(d, prior)[0], dev.build(d, prior)[0]), (levels, t)
            assert dev.last_on_gpu
    host.close()
    dev.close()


@pytest.mark.parametrize("tag,opts", [("", []), ("_nc", ["--no_consistency"])])
def test_cli_sample_ages_on_the_device_builder(tmp_path, tag, opts):
    """the stage with --sample_ages and the device builder: the reference's .anc / .mut for the same ages file
    (tests/golden/synth24_ages.npz), every tree built by the device's workers"""
    from golden_util import Fixture
    cli = os.path.join(os.path.dirname(os.path.abspath(api.__file__)), "Relate")
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth24_ages", work / "out")
    with open(work / "ages.txt", "w") as f:
        f.write("\n".join("%g" % a for a in fx.z["ages"]) + "\n")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    env = dict(os.environ, RELATE_AMD_GPU_BUILD="1", RELATE_AMD_TIMING="1")
    p = subprocess.run([cli, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                        str(fx.W - 1), "--sample_ages", "ages.txt", "-o", "out"] + opts, cwd=str(work),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut%s/%d" % (tag, w)].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc%s/%d" % (tag, w)].tobytes(), w
    counts = re.findall(r"(\d+) trees on the GPU, (\d+) on the host", p.stderr.decode())
    assert counts and sum(int(g) for g, _ in counts) > 0 and all(int(h) == 0 for _, h in counts), counts


def test_stage_on_a_panel_with_ancient_samples(tmp_path):
    """Paint + BuildTopology of a synthetic chunk (N = 1000, 1600 SNPs, several sections) with a tenth of the samples
    ancient: the device builder's files against the host builder's -- the one held to the reference's -- byte for byte,
    and (nearly) every tree built by the device's workers"""
    import rlutil
    from bigtile import link_inputs
    cli = os.path.join(os.path.dirname(os.path.abspath(api.__file__)), "Relate")
    N, L = 1000, 1600
    ch = rlutil.synth_chunk(N, L, seed=31, budget=3e7)
    assert ch.W >= 3
    rng = np.random.RandomState(9)
    ages = np.zeros(N)
    ages[rng.rand(N) < 0.1] = 800.0
    ages[rng.rand(N) < 0.04] = 2400.0
    ages = np.repeat(ages[::2], 2)  # (the two haplotypes of a sample share its age)
    with open(tmp_path / "ages.txt", "w") as f:
        f.write("\n".join("%g" % a for a in ages) + "\n")
    ch.write(str(tmp_path / "host" / "out"))
    link_inputs(str(tmp_path / "host" / "out"), str(tmp_path / "dev" / "out"))
    counts = {}
    for which, gpu in (("host", "0"), ("dev", "1")):
        env = dict(os.environ, RELATE_AMD_GPU_BUILD=gpu, RELATE_AMD_TIMING="1")
        p = subprocess.run([cli, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--sample_ages", "../ages.txt", "-o",
                            "out"], cwd=str(tmp_path / which), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        found = re.findall(r"(\d+) trees on the GPU, (\d+) on the host", p.stderr.decode())
        counts[which] = (sum(int(g) for g, _ in found), sum(int(h) for _, h in found))
    for w in range(ch.W):
        for ext in ("anc", "mut"):
            a = open(tmp_path / "host" / "out" / "chunk_0" / ("out_%d.%s" % (w, ext)), "rb").read()
            b = open(tmp_path / "dev" / "out" / "chunk_0" / ("out_%d.%s" % (w, ext)), "rb").read()
            assert a == b, (w, ext)
    assert counts["host"][0] == 0 and counts["dev"][0] > 0
    assert counts["dev"][1] <= counts["dev"][0] // 20, counts
    from bigtile import record
    record("ages_stage_N1000", {"trees_on_device": counts["dev"][0], "trees_on_host": counts["dev"][1]})
