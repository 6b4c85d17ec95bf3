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
