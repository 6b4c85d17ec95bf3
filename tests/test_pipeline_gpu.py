"""The many-chunks route end to end on the GPU box (BASELINE.json config #4's shape; scripts/RelateParallel/
RelateParallel.sh:216-262): a synthetic .haps -> this library's MakeChunks (3 overlapping chunks, each with its
bit-packed panel chunk_<i>.bits) -> relate_amd.dist.run_chunks (one rank: every chunk through the fused Paint +
BuildTopology stage, then FindEquivalentBranches) -- every .anc / .mut against what the REFERENCE binary wrote for
the same job (tests/golden/pipeline6.npz, tools/make_golden.py pipeline).  The paint files of the two-stage route
likewise, from the bit-packed panel and from the reference's char panel."""
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

from relate_amd import api
from relate_amd import dist as rdist
from test_makechunks import write_synth_haps

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def md5(path):
    return np.frombuffer(hashlib.md5(open(path, "rb").read()).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def job(tmp_path_factory):
    z = np.load(os.path.join(ROOT, "tests", "golden", "pipeline6.npz"))
    N, L = [int(x) for x in z["args"]]
    work = str(tmp_path_factory.mktemp("pipeline"))
    write_synth_haps(work, N, L, seed=N)
    for fn in ("s.haps", "s.sample", "s.map"):
        assert np.array_equal(md5(os.path.join(work, fn)), z["in_md5/" + fn]), fn
    p = subprocess.run([CLI, "--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map",
                        "--memory", "%g" % float(z["memory"][0]), "-o", "job"], cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_CHUNK_BITS="1"))  # (the bit-packed panel next to the .hap: opt-in)
    assert p.returncode == 0, p.stderr.decode()
    C = int(z["num_chunks"][0])
    assert rdist.read_parameters(os.path.join(work, "job"))["num_chunks"] == C and C >= 2
    shutil.copytree(os.path.join(work, "job"), os.path.join(work, "pristine"))  # (copy2: modification times kept)
    return z, work, C


def check_trees(z, out, c):
    W = int(z["c%d/sections" % c][0])
    assert api.num_sections(out, c) == W
    for w in range(W):
        for ext in ("anc", "mut"):
            assert np.array_equal(md5(os.path.join(out, "chunk_%d" % c, "job_%d.%s" % (w, ext))),
                                  z["c%d/job_%d.%s" % (c, w, ext)]), (c, w, ext)


def test_every_chunk_through_run_chunks(job):
    z, work, C = job
    out = os.path.join(work, "job")
    assert rdist.run_chunks(out) == list(range(C))  # (one rank: all chunks; the fused stage, no paint files)
    for c in range(C):
        assert not os.path.exists(os.path.join(out, "chunk_%d" % c, "paint"))
        check_trees(z, out, c)
        shutil.rmtree(os.path.join(out, "chunk_%d" % c))


@pytest.mark.parametrize("panel", ["bits", "hap", "stale_bits"])
def test_two_stage_route_from_either_panel(job, panel):
    """Paint -> paint files -> BuildTopology -> FindEquivalentBranches (run_chunks(paint_files=True)); with
    chunk_<i>.bits present rl_load_chunk reads the bit-packed panel, without it -- or next to a .hap it was not written
    for (another size or modification time) -- the reference's char panel; the last stage removes the .bits"""
    z, work, C = job
    out = os.path.join(work, "job_" + panel)
    shutil.copytree(os.path.join(work, "pristine"), out)
    bits = [f for f in os.listdir(out) if f.endswith(".bits")]
    assert len(bits) == C
    if panel == "hap":
        for f in bits:
            os.remove(os.path.join(out, f))
    elif panel == "stale_bits":  # a .hap written after the .bits: the .bits (made useless here) must not be read
        for c in range(C):
            with open(os.path.join(out, "chunk_%d.bits" % c), "r+b") as f:
                f.seek(32)
                f.write(b"\xff" * 64)
            st = os.stat(os.path.join(out, "chunk_%d.hap" % c))
            os.utime(os.path.join(out, "chunk_%d.hap" % c), ns=(st.st_atime_ns, st.st_mtime_ns + 1000000000))
    else:  # the char panel must not be what is read: break it (same size, same modification time)
        for c in range(C):
            fn = os.path.join(out, "chunk_%d.hap" % c)
            st = os.stat(fn)
            with open(fn, "r+b") as f:
                f.seek(16)
                f.write(b"\x00" * 64)
            os.utime(fn, ns=(st.st_atime_ns, st.st_mtime_ns))
    # (the output name is the directory's base name, as with the CLI's -o)
    os.rename(out, os.path.join(work, "tmp_" + panel))
    os.makedirs(os.path.join(work, panel))
    out = os.path.join(work, panel, "job")
    os.rename(os.path.join(work, "tmp_" + panel), out)
    assert rdist.run_chunks(out, paint_files=True, chunks=[0, C - 1]) == [0, C - 1]
    for c in (0, C - 1):
        W = int(z["c%d/sections" % c][0])
        for w in range(W):
            assert np.array_equal(md5(os.path.join(out, "chunk_%d" % c, "paint", "relate_%d.bin" % w)),
                                  z["c%d/paint/relate_%d.bin" % (c, w)]), (c, w)
        check_trees(z, out, c)
        assert not os.path.exists(os.path.join(out, "chunk_%d.bits" % c))  # (FindEquivalentBranches took it along)
