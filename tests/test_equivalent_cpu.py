"""FindEquivalentBranches (the stage after BuildTopology, host code) against the
reference binary: the .anc files it rewrites are byte-identical to the
reference's on every fixture (tests/golden/*.npz, `feb_anc/*` made by
tools/make_golden.py from the unmodified reference)."""
import os
import subprocess

import pytest

from golden_util import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


@pytest.mark.parametrize("name", ["synth24", "synth70", "example8", "synth40_noisy"])
def test_find_equivalent_branches_matches_reference(tmp_path, name):
    out = tmp_path / "out"
    out.mkdir()
    fx = Fixture(name, out)
    cdir = out / "chunk_0"
    cdir.mkdir()
    for w in range(fx.W):  # the BuildTopology outputs of the reference
        (cdir / ("out_%d.anc" % w)).write_bytes(fx.z["anc/%d" % w].tobytes())
    p = subprocess.run([CLI, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path),
                       stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert b"Propagating mutations across AncesTrees" in p.stderr and b"CPU Time spent" in p.stderr
    changed = 0
    for w in range(fx.W):
        got = (cdir / ("out_%d.anc" % w)).read_bytes()
        assert got == fx.z["feb_anc/%d" % w].tobytes(), "window %d" % w
        changed += got != fx.z["anc/%d" % w].tobytes()
    assert changed > 0  # the stage did something
    assert not [f for f in os.listdir(cdir) if f.startswith("equivalent_branches")]


def test_find_equivalent_branches_reports_missing_files(tmp_path):
    out = tmp_path / "out"
    out.mkdir()
    Fixture("synth24", out)
    p = subprocess.run([CLI, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"], cwd=str(tmp_path),
                       stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"cannot open" in p.stderr


@pytest.mark.parametrize("name,pool,order", [("synth24", 3, "reversed"), ("synth70", 1, "shuffled"),
                                             ("synth40_noisy", 8, "shuffled"), ("synth24", 2, "in order")])
def test_the_fused_job_from_sections_in_any_order(tmp_path, name, pool, order):
    """rl_stage_opts.find_equivalent_branches: the association fused behind BuildTopology (equivalent.cpp: FebJob) --
    sections handed over as they finish (any order), their neighbouring trees associated on pool threads, the pairs
    across section boundaries at the end, every .anc written once.  Here fed from the reference's BuildTopology files
    through a test hook (no GPU): the bytes the reference's FindEquivalentBranches leaves."""
    import ctypes as C
    import numpy as np
    from relate_amd import api
    out = tmp_path / "out"
    out.mkdir()
    fx = Fixture(name, out)
    cdir = out / "chunk_0"
    cdir.mkdir()
    for w in range(fx.W):
        (cdir / ("out_%d.anc" % w)).write_bytes(fx.z["anc/%d" % w].tobytes())
    secs = list(range(fx.W))
    if order == "reversed":
        secs.reverse()
    elif order == "shuffled":
        np.random.RandomState(fx.W).shuffle(secs)
    arr = np.array(secs, dtype=np.int32)
    lib = api.lib()
    lib.rl_debug_feb_fused_from_files.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    rc = lib.rl_debug_feb_fused_from_files(str(out).encode(), 0, arr.ctypes.data_as(C.c_void_p), len(secs), pool)
    assert rc == 0, lib.rl_last_error()
    for w in range(fx.W):
        assert (cdir / ("out_%d.anc" % w)).read_bytes() == fx.z["feb_anc/%d" % w].tobytes(), "window %d" % w
