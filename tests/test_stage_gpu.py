"""The drop-in CLI (relate_amd/Relate --mode Paint / --mode BuildTopology) on
the GPU against the committed outputs of the reference binary: every paint
file, .anc and .mut byte-identical."""
import os
import subprocess

import pytest

from golden_util import Fixture

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def run_cli(args, cwd):
    p = subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return p.stderr.decode()


@pytest.mark.parametrize("name", ["synth24", "synth70", "example8", "synth40_noisy"])
def test_cli_stages_byte_identical_to_reference(tmp_path, name):
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture(name, work / "out")
    err = run_cli(["--mode", "Paint", "--chunk_index", "0", "-o", "out"], str(work))
    assert "CPU Time spent" in err
    for w in range(fx.W):
        got = open(work / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w), "rb").read()
        assert got == fx.paint_file(w), "paint window %d" % w
    run_cli(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
             str(fx.W - 1), "-o", "out"], str(work))
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


def test_cli_rejects_unknown_option_and_other_modes(tmp_path):
    p = subprocess.run([CLI, "--mode", "Paint", "--bogus", "1", "-o", "out", "--chunk_index", "0"], cwd=str(tmp_path),
                       stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"does not exist" in p.stderr
    p = subprocess.run([CLI, "--mode", "Finalize", "-o", "out", "--chunk_index", "0"], cwd=str(tmp_path),
                       stderr=subprocess.PIPE)
    assert p.returncode != 0
    p = subprocess.run([CLI, "--mode", "Paint", "-o", "a/b", "--chunk_index", "0"], cwd=str(tmp_path),
                       stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"working directory" in p.stderr


def test_chunk_pipeline_through_python(tmp_path):
    """relate_amd.dist.run_chunk (one rank): Paint -> BuildTopology -> FindEquivalentBranches, every file as the
    reference leaves it"""
    from relate_amd import dist as rdist
    out = tmp_path / "out"
    out.mkdir()
    fx = Fixture("synth70", out)
    assert rdist.run_chunk(str(out), 0) == (0, fx.W - 1)
    for w in range(fx.W):
        assert open(out / "chunk_0" / "paint" / ("relate_%d.bin" % w), "rb").read() == fx.paint_file(w), w
        assert open(out / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(out / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["feb_anc/%d" % w].tobytes(), w


@pytest.mark.parametrize("tag,opts", [("nc", ["--no_consistency"]), ("fb", ["--fb", "2500"])])
def test_cli_build_topology_options(tmp_path, tag, opts):
    """--no_consistency / --fb through the CLI: .anc and .mut as the reference writes them with the same option"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    run_cli(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
             str(fx.W - 1), "-o", "out"] + opts, str(work))
    differs = 0
    for w in range(fx.W):
        mut = open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read()
        anc = open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read()
        assert mut == fx.z["mut_%s/%d" % (tag, w)].tobytes(), w
        assert anc == fx.z["anc_%s/%d" % (tag, w)].tobytes(), w
        differs += anc != fx.z["anc/%d" % w].tobytes()
    assert differs > 0  # the option changed something


def test_cli_build_topology_bounded_windows(tmp_path):
    """windows that keep a fraction of their posterior rows resident (as the stage does by itself when the open
    sections would not fit in HBM): the same .anc / .mut bytes"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(fx.W - 1), "-o", "out"], cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_WINDOW_ROWS="300"))
    assert p.returncode == 0, p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


@pytest.mark.parametrize("name", ["synth24", "synth70", "example8", "synth40_noisy"])
def test_cli_build_topology_trees_built_on_the_gpu(tmp_path, name):
    """RELATE_AMD_GPU_BUILD=1: MinMatch itself on the GPU (one workgroup per tree), host builder as the fallback
    for trees that need the symmetric matrix -- the same .anc / .mut bytes as the reference"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture(name, work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(fx.W - 1), "-o", "out"], cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_GPU_BUILD="1", RELATE_AMD_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()
    import re
    on_gpu = sum(int(x) for x in re.findall(r"(\d+) trees on the GPU", p.stderr.decode()))
    on_host = sum(int(x) for x in re.findall(r"trees on the GPU, (\d+) on the host", p.stderr.decode()))
    assert on_gpu > 0 and on_host == 0
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


def test_cli_gpu_build_with_bounded_windows(tmp_path):
    """both together: windows that repaint in parts feed the device-resident tree builder"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(fx.W - 1), "-o", "out"], cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_GPU_BUILD="1", RELATE_AMD_WINDOW_ROWS="300"))
    assert p.returncode == 0, p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


def test_fused_stage_with_two_repaint_lanes(tmp_path):
    """RELATE_AMD_REPAINT_LANES=2: the windows of the sections repaint on two streams with strips of their own, side by
    side (bounded windows, so that every section comes back for more launches): the same bytes"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    p = subprocess.run([CLI, "--mode", "PaintBuildTopology", "--chunk_index", "0", "-o", "out"], cwd=str(work),
                       stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_REPAINT_LANES="2", RELATE_AMD_WINDOW_ROWS="300",
                                RELATE_AMD_GPU_BUILD="1"))
    assert p.returncode == 0, p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


@pytest.mark.parametrize("tag,opts", [("nc", ["--no_consistency"]), ("fb", ["--fb", "2500"])])
def test_cli_gpu_build_options(tmp_path, tag, opts):
    """--no_consistency (no penalty, no prior) and --fb with the trees built on the GPU"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(fx.W - 1), "-o", "out"] + opts, cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_GPU_BUILD="1"))
    assert p.returncode == 0, p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut_%s/%d" % (tag, w)].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc_%s/%d" % (tag, w)].tobytes(), w


@pytest.mark.parametrize("name", ["synth24", "synth70", "example8", "synth40_noisy"])
def test_paint_and_build_topology_without_paint_files(tmp_path, name):
    """--mode PaintBuildTopology (rl_stage_paint_build_topology): the stepping stones stay in HBM, the paint file's
    float / run-length quantisation is applied on the device -- no chunk_0/paint directory, and the same .anc / .mut
    as the reference's Paint + BuildTopology through its paint files (Relate.cpp:257-283)"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture(name, work / "out")
    run_cli(["--mode", "PaintBuildTopology", "--chunk_index", "0", "-o", "out"], str(work))
    assert not (work / "out" / "chunk_0" / "paint").exists()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


def test_fused_stage_with_the_stones_parked_on_the_host(tmp_path):
    """the fused stage of a chunk whose stones would crowd out the sections' windows (C3: 53 GB) moves them to pinned
    host memory after Paint and every window takes its slice back (rl_park_stones; RELATE_AMD_PARK_STONES=1 forces
    it at this size): the same bytes"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    p = subprocess.run([CLI, "--mode", "PaintBuildTopology", "--chunk_index", "0", "-o", "out"], cwd=str(work),
                       stderr=subprocess.PIPE, env=dict(os.environ, RELATE_AMD_PARK_STONES="1"))
    assert p.returncode == 0, p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w


def test_in_memory_window_matches_the_paint_file_window(tmp_path):
    """a window opened from the context's stones (quantised on the device) against the same window opened from the
    paint file the context wrote: posterior rows and matrices bit for bit"""
    import numpy as np
    import rlutil
    from relate_amd import api
    ch = rlutil.synth_chunk(200, 1500, seed=7, budget=400000)
    ctx = api.Context()
    ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    ctx.paint(api.RL_SUM_EXACT)
    ctx.write_paint_files(str(tmp_path))
    for w in sorted(set([0, ch.W // 2, ch.W - 1])):
        s0 = int(ch.wb[w])
        a = ctx.open_window(w, os.path.join(str(tmp_path), "relate_%d.bin" % w), s0)
        b = ctx.open_window(w, None, s0)
        for n in (0, 77, 199):
            ta, la = a.topology(n)
            tb, lb = b.topology(n)
            assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32)) and np.array_equal(la.view(np.uint32), lb.view(np.uint32))
        assert np.array_equal(a.matrix(s0).view(np.uint32), b.matrix(s0).view(np.uint32))
        a.close()
        b.close()
    ctx.close()


@pytest.mark.parametrize("tag,opts", [("", []), ("_nc", ["--no_consistency"])])
def test_cli_build_topology_with_sample_ages(tmp_path, tag, opts):
    """--sample_ages (ancient samples; pipeline/BuildTopology.cpp:93-108, the third candidate key and the clock of
    tree_builder.cpp): .anc / .mut of every section as the reference writes them for the same ages file
    (tests/golden/synth24_ages.npz, tools/make_golden.py ages); and the ages do change the trees"""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth24_ages", work / "out")
    with open(work / "ages.txt", "w") as f:
        f.write("\n".join("%g" % a for a in fx.z["ages"]) + "\n")
    paint = work / "out" / "chunk_0" / "paint"
    fx.write_paint_files(str(paint))
    run_cli(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section", str(fx.W - 1),
             "--sample_ages", "ages.txt", "-o", "out"] + opts, str(work))
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut%s/%d" % (tag, w)].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc%s/%d" % (tag, w)].tobytes(), w
    if not tag:
        plain = Fixture("synth24", tmp_path / "plain") if (tmp_path / "plain").mkdir() is None else None
        assert any(fx.z["anc/%d" % w].tobytes() != plain.z["anc/%d" % w].tobytes() for w in range(fx.W))


def test_stage_options_through_the_abi(tmp_path):
    """rl_stage_opts (include/relate_amd.h): what the tests elsewhere set through the environment, per call -- two calls
    of one process with different options: (1) sample ages through `sample_ages_path` (not the process-wide setter);
    (2) no ages, the device builder, windows bounded to a third of their rows, one section thread, on the
    plain fixture: each byte-identical to the reference's files for ITS options"""
    from relate_amd import api
    ages_dir, plain_dir = tmp_path / "ages" / "out", tmp_path / "plain" / "out"
    ages_dir.mkdir(parents=True)
    plain_dir.mkdir(parents=True)
    fa, fp = Fixture("synth24_ages", ages_dir), Fixture("synth24", plain_dir)
    with open(tmp_path / "ages.txt", "w") as f:
        f.write("\n".join("%g" % a for a in fa.z["ages"]) + "\n")
    fa.write_paint_files(str(ages_dir / "chunk_0" / "paint"))
    api.stage_build_topology_ex(str(ages_dir), 0, 0, fa.W - 1, api.stage_opts(sample_ages_path=str(tmp_path / "ages.txt")))
    for w in range(fa.W):
        assert open(ages_dir / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fa.z["anc/%d" % w].tobytes(), w
        assert open(ages_dir / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fa.z["mut/%d" % w].tobytes(), w
    # the fused stage next, in the same process: the first call's ages must not stick
    ctx = api.Context()
    ctx.load_chunk(str(plain_dir), 0)
    ctx.paint()
    rows = max(sum(ctx.open_window(w, None, int(fp.chunk.wb[w])).rows(n) for n in range(fp.N)) for w in range(fp.W))
    ctx.close()
    api.stage_build_topology_ex(str(plain_dir), 0, 0, fp.W - 1,
                                api.stage_opts(gpu_build=1, window_rows=max(8, rows // 3), section_threads=1, flags=1),
                                fused=True)
    for w in range(fp.W):
        assert open(plain_dir / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fp.z["anc_nc/%d" % w].tobytes(), w
        assert open(plain_dir / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fp.z["mut_nc/%d" % w].tobytes(), w


def test_cli_fast_mode_through_the_file_stages(tmp_path):
    """--sum_mode lanes32 through the two file-based stages: Paint writes paint files of the reference's format (every
    record decodes; the same boundary SNPs as the reference's), BuildTopology builds valid binary trees from them at the
    reference's tree positions or others -- the fast mode promises the tolerance on the distances, not the bytes"""
    import numpy as np
    import rlutil
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth70", work / "out")
    run_cli(["--mode", "Paint", "--chunk_index", "0", "--sum_mode", "lanes32", "-o", "out"], str(work))
    for w in range(fx.W):
        got = open(work / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w), "rb").read()
        assert got[:8] == fx.paint_file(w)[:8]  # (section_startpos, section_endpos of the first record)
        assert 0.5 < len(got) / len(fx.paint_file(w)) < 2.0
    run_cli(["--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section", str(fx.W - 1),
             "--sum_mode", "lanes32", "-o", "out"], str(work))
    for w in range(fx.W):
        N, trees = rlutil.parse_anc(str(work / "out" / "chunk_0" / ("out_%d.anc" % w)))
        assert N == fx.N and len(trees) >= 1
        for t in trees:
            par = t[1]
            assert par[-1] == -1 and np.all(np.bincount(par[:-1], minlength=2 * N - 1)[N:] == 2)


@pytest.mark.parametrize("name,mode,builder", [("synth24", "PaintBuildTopology", "1"), ("synth70", "PaintBuildTopology", "0"),
                                               ("synth40_noisy", "BuildTopology", "1"), ("example8", "PaintBuildTopology", "1")])
def test_find_equivalent_branches_fused_behind_build_topology(tmp_path, name, mode, builder):
    """--find_equivalent_branches (rl_stage_opts.find_equivalent_branches): the stage downstream
    (pipeline/FindEquivalentBranches.cpp:13-167) runs on the sections' trees while they are in memory and every .anc is
    written once -- the bytes the reference's BuildTopology + FindEquivalentBranches leave (`feb_anc/*` of the
    fixtures), the .mut files untouched; through both stage entry points, host and device builder.  A call that does
    not cover the whole chunk is refused."""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture(name, work / "out")
    if "feb_anc/0" not in fx.z.files or fx.W < 2:
        pytest.skip("fixture without FindEquivalentBranches outputs / with one section")
    if mode == "BuildTopology":
        run_cli(["--mode", "Paint", "--chunk_index", "0", "-o", "out"], str(work))
    args = ["--mode", mode, "--chunk_index", "0", "--first_section", "0", "--last_section", str(fx.W - 1), "-o", "out",
            "--find_equivalent_branches"]
    p = subprocess.run([CLI] + args, cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_GPU_BUILD=builder, RELATE_AMD_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()
    assert "find equivalent branches, fused" in p.stderr.decode()
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["feb_anc/%d" % w].tobytes(), w
    p = subprocess.run([CLI, "--mode", mode, "--chunk_index", "0", "--first_section", "0", "--last_section", "0", "-o",
                        "out", "--find_equivalent_branches"], cwd=str(work), stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"covers all" in p.stderr


def test_an_out_of_memory_window_open_waits_instead_of_failing_the_stage(tmp_path):
    """a window that is admitted on the byte count but finds no block to take must go back to waiting while
    other sections hold memory they will give back -- not fail the stage.  RELATE_AMD_TEST_FAIL_OPENS makes admitted
    opens fail as if the allocator had run dry: with other sections open the stage waits and finishes with the
    reference's bytes; when the ONLY section's open fails there is nobody to wait for and the stage reports it."""
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture("synth24", work / "out")
    env = dict(os.environ, RELATE_AMD_GPU_BUILD="1", RELATE_AMD_SECTION_THREADS="4", RELATE_AMD_TEST_FAIL_OPENS="3")
    # (three opens that happen while other sections are open fail: those threads wait for a section to close and retry)
    p = subprocess.run([CLI, "--mode", "PaintBuildTopology", "--chunk_index", "0", "-o", "out"], cwd=str(work),
                       stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w
    p = subprocess.run([CLI, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", "0", "-o", "out"], cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_TEST_FAIL_OPENS="-3"))  # (the open and its two retries)
    assert p.returncode != 0 and b"hipMalloc failed" in p.stderr
