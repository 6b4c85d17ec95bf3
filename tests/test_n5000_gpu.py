"""The headline tile against the reference binary: N = 5000 haplotypes (BASELINE.json config #3's N: one wavefront
of S = 80 registers per target), a chunk short enough for the single-threaded reference to finish in minutes
(L = 1200, 2 windows).  tests/golden/n5000.npz (tools/make_golden.py n5000) holds the reference's md5 of every
output file of Paint (whole chunk) and BuildTopology (section 0), the head of window 0's paint file, the .mut file
and the parent arrays (md5 per tree, three in full); the inputs are regenerated from the seed and checked against
their md5s.  Through the drop-in CLI, with the trees built on the host and on the GPU; also the shipped variants of
the path at this tile: the fused stage (no paint files, stones quantised on the device), bounded windows (a part of
the posterior rows resident, RePaint again as the builder moves on) and the `lanes` summation order against the
reference's distance matrices (tests/golden/n5000_matrix.npz, tools/make_golden.py n5000_matrix)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import rlutil
from relate_amd import api

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")
GOLD = os.path.join(ROOT, "tests", "golden", "n5000.npz")


from bigtile import md5, check_section_0, run_cli  # noqa: E402
import bigtile  # noqa: E402


@pytest.fixture(scope="module")
def painted(tmp_path_factory):
    work = str(tmp_path_factory.mktemp("n5000"))
    z, W = bigtile.make_chunk_dir(GOLD, work)
    run_cli("Paint", work, "host")
    return z, work, W


def test_paint_files_of_the_headline_tile(painted):
    z, work, W = painted
    ctx = api.Context()
    ctx.load_chunk(os.path.join(work, "out"), 0)
    assert (ctx.N, ctx.tile, ctx.waves) == (5000, 80, 1)
    ctx.close()
    for w in range(W):
        b = open(os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % w), "rb").read()
        if w == 0:
            head = z["head/paint/relate_0.bin"].tobytes()
            assert b[:len(head)] == head, "window 0's paint file differs within its first %d bytes" % len(head)
        assert np.array_equal(md5(b), z["md5/paint/relate_%d.bin" % w]), "paint file of window %d" % w


@pytest.mark.parametrize("builder", ["host", "gpu"])
def test_section_0_trees_of_the_headline_tile(painted, builder):
    z, work, W = painted
    run_cli("BuildTopology", work, builder)
    check_section_0(z, os.path.join(work, "out"))


def window0_rows(painted):
    z, work, W = painted
    ctx = api.Context()
    ctx.load_chunk(os.path.join(work, "out"), 0)
    win = ctx.open_window(0, os.path.join(work, "out", "chunk_0", "paint", "relate_0.bin"), int(z["wb"][0]))
    rows = sum(win.rows(n) for n in range(ctx.N))
    win.close()
    ctx.close()
    return rows


def repaint_launches(stderr_text):
    import re
    m = re.search(r"(\d+) RePaint launches", stderr_text)
    return int(m.group(1)) if m else -1


@pytest.mark.parametrize("builder", ["host", "gpu"])
def test_bounded_windows_at_the_headline_tile(painted, builder):
    """a fifth of window 0's posterior rows resident (what the stage does at C3, where 150 sections share HBM):
    RePaint runs again as the builder moves on -- same trees, same bytes"""
    z, work, W = painted
    err = run_cli("BuildTopology", work, builder, {"RELATE_AMD_WINDOW_ROWS": str(window0_rows(painted) // 5)})
    assert repaint_launches(err) >= 4, err[-600:]
    check_section_0(z, os.path.join(work, "out"))


@pytest.mark.parametrize("builder,bounded", [("host", False), ("gpu", False), ("gpu", True)])
def test_fused_stage_at_the_headline_tile(painted, tmp_path, builder, bounded):
    """--mode PaintBuildTopology: no paint files, the stepping stones stay in HBM and take the paint file's
    quantisation on the device -- the reference's .anc / .mut of section 0"""
    z, work, W = painted
    out = tmp_path / "out"
    out.mkdir()
    for f in os.listdir(os.path.join(work, "out")):
        if os.path.isfile(os.path.join(work, "out", f)):
            os.symlink(os.path.join(work, "out", f), str(out / f))
    env = {"RELATE_AMD_WINDOW_ROWS": str(window0_rows(painted) // 5)} if bounded else {}
    err = run_cli("PaintBuildTopology", str(tmp_path), builder, env)
    assert not os.path.exists(str(out / "chunk_0" / "paint"))
    assert repaint_launches(err) >= (4 if bounded else 1), err[-600:]
    check_section_0(z, str(out))


def test_lanes_mode_at_the_headline_tile(painted):
    """RL_SUM_LANES (the fast, re-associated normalising sums) against the REFERENCE's distance matrices at N = 5000:
    |d_lanes - d_ref| <= 1e-5 * max(|d|, max |logscale|), and how many entries / trees stay identical;
    the exact mode on the same rows must be bit-identical"""
    z, work, W = painted
    zm = np.load(os.path.join(ROOT, "tests", "golden", "n5000_matrix.npz"))
    assert [int(x) for x in zm["meta"]] == [int(x) for x in z["meta"]]
    rows = [int(x) for x in zm["rows"]]
    scale = max(1.0, float(zm["logscale_max"][0]))
    report = {}
    for mode, name in ((api.RL_SUM_EXACT, "exact"), (api.RL_SUM_LANES, "lanes"), (api.RL_SUM_LANES32, "lanes32")):
        ctx = api.Context()
        ctx.load_chunk(os.path.join(work, "out"), 0)
        ctx.paint(mode)
        win = ctx.open_window(0, None, int(zm["snps"][0]), mode)
        cur = int(zm["snps"][0])
        worst, same = 0.0, []
        for i, s in enumerate(int(x) for x in zm["snps"]):
            for t in range(cur + 1, s + 1):
                win.advance(t)
            cur = s
            g = win.matrix(s)
            ref = zm["matrix_rows/%d" % i]
            if mode == api.RL_SUM_EXACT:
                assert np.array_equal(md5(np.ascontiguousarray(g).tobytes()), zm["matrix_md5/%d" % i]), s
                assert np.array_equal(g[rows].view(np.uint32), ref.view(np.uint32)), s
            else:
                tol = 1e-5 * np.maximum(np.abs(ref), scale)
                diff = np.abs(g[rows] - ref)
                assert np.all(diff <= tol), (s, float(diff.max()), scale)
                worst = max(worst, float((diff / tol).max()))
                same.append(float(np.mean(g[rows].view(np.uint32) == ref.view(np.uint32))))
        report[name] = (worst, same)
        win.close()
        ctx.close()
    for name in ("lanes", "lanes32"):
        print("%s vs reference at N=5000: worst |diff| / tolerance %.3f, identical entries per matrix %s"
              % (name, report[name][0], ["%.4f" % x for x in report[name][1]]))
        bigtile.record("n5000_" + name + "_matrices", {"worst_diff_over_tolerance": report[name][0],
                                                        "identical_entries_per_matrix": report[name][1]})
        assert report[name][0] <= 1.0


@pytest.mark.parametrize("mode", ["lanes", "lanes32"])
def test_lanes_mode_trees_at_the_headline_tile(painted, tmp_path, mode):
    """the trees `lanes` / `lanes32` build from their own stones: a valid tree sequence at the reference's positions is NOT
    promised (MinMatch breaks exact float ties) -- the fraction of identical parent arrays is
    reported, the run must succeed and produce binary trees"""
    z, work, W = painted
    out = tmp_path / "out"
    out.mkdir()
    for f in os.listdir(os.path.join(work, "out")):
        if os.path.isfile(os.path.join(work, "out", f)):
            os.symlink(os.path.join(work, "out", f), str(out / f))
    env = dict(os.environ, RELATE_AMD_GPU_BUILD="1")
    p = subprocess.run([CLI, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", "0", "--sum_mode", mode, "-o", "out"], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    N, trees = rlutil.parse_anc(str(out / "chunk_0" / "out_0.anc"))
    ref_pos = list(z["tree_pos"])
    same = 0
    for t in trees:
        par = t[1]
        assert par[-1] == -1 and np.all(np.bincount(par[:-1], minlength=2 * N - 1)[N:] == 2)
        if t[0] in ref_pos and np.array_equal(md5(par.astype("<i4").tobytes()), z["tree_parent_md5"][ref_pos.index(t[0])]):
            same += 1
    print("%s at N=5000, section 0: %d trees (reference %d), %d with the reference's parent array (%.1f %%)"
          % (mode, len(trees), len(ref_pos), same, 100.0 * same / max(1, len(ref_pos))))
    bigtile.record("n5000_" + mode + "_trees", {"trees": len(trees), "reference_trees": len(ref_pos),
                                               "identical_parent_arrays": same,
                                               "identical_fraction": same / max(1, len(ref_pos))})
    # A floor, so that a regression is red.  `lanes` (same arithmetic on doubles, sums re-associated): the stones are
    # floats behind a 1e-3 run-length quantisation, which swallows its 1e-16 -- every tree of this section is the
    # reference's, though nothing promises that.  `lanes32` moves 15-25 % of the distances in their last bits (1e-7
    # relative, 0.02 of the tolerance) and MinMatch's merges hinge on exact float comparisons: one flipped merge among
    # 4999 makes another parent array, so almost every tree differs (8 of 155 identical when this was written) -- as
    # valid as the trees the reference builds from a panel with one allele changed, but not THE trees.  No floor.
    if mode == "lanes":
        assert same >= 0.5 * len(ref_pos)

