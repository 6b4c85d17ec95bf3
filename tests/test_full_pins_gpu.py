"""BASELINE.json configs #2 and #4 against the unmodified reference at FULL length.

tests/golden/full_c2.npz / full_c4.npz (tools/make_golden_full.py, the reference binary in the build container):
  * config #2: synthetic N = 1000 x L = 100,000, seed 1, --memory 5 (11 windows) -- the chunk bench.py paints for C2;
  * one chunk of config #4 as bench.py makes it: N = 2000 x L = 121,000, seed 1, --memory 1;
for each the reference's `Relate --mode Paint` of the WHOLE chunk (md5 + size of every window's paint file,
pipeline/Paint.cpp:17-108 -> fast_painting.cpp:18-618) and `--mode BuildTopology` of the first, a middle and the last
section (pipeline/BuildTopology.cpp:125-150): md5 of .anc / .mut, the .mut in full, every tree's position and the md5
of its parent array.

Here, through the drop-in CLI on the same chunk files (md5-checked): Paint -> every paint file byte-identical;
BuildTopology of those sections from the files; and the fused PaintBuildTopology stage of ALL sections (the device
tree builder, stones in HBM) -> the same three sections' files again."""
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


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return np.frombuffer(h.digest(), dtype=np.uint8)


def md5(b):
    return np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)


@pytest.fixture(scope="module", params=["full_c2", "full_c4"])
def pinned(request, tmp_path_factory):
    z = np.load(os.path.join(ROOT, "tests", "golden", request.param + ".npz"))
    N, L, W, seed = [int(x) for x in z["meta"]]
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = float(z["mem"][0]) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499) == W
    assert np.array_equal(wb[:W + 1], z["wb"])
    work = str(tmp_path_factory.mktemp(request.param))
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    for key in z.files:  # the chunk files the reference was given
        if key.startswith("in_md5/"):
            assert np.array_equal(md5_file(os.path.join(d, key[7:])), z[key]), key
    return z, work, (N, L, W)


def run(work, *args, env=None):
    p = subprocess.run([CLI] + list(args) + ["--chunk_index", "0", "-o", "out"], cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-600:]


def check_sections(z, work, keep_paint=False):
    d = os.path.join(work, "out", "chunk_0")
    for s in (int(x) for x in z["sections"]):
        anc = os.path.join(d, "out_%d.anc" % s)
        mut = open(os.path.join(d, "out_%d.mut" % s), "rb").read()
        _, trees = rlutil.parse_anc(anc)
        assert [t[0] for t in trees] == list(z["s%d/tree_pos" % s]), "tree positions of section %d" % s
        for t, (tr, want) in enumerate(zip(trees, z["s%d/tree_parent_md5" % s])):
            assert np.array_equal(md5(tr[1].astype("<i4").tobytes()), want), "section %d, parent array of tree %d" % (s, t)
        assert mut == z["s%d/mut" % s].tobytes(), "section %d: .mut" % s
        assert os.path.getsize(anc) == int(z["s%d/anc_size" % s][0])
        assert np.array_equal(md5_file(anc), z["s%d/anc_md5" % s]), "section %d: .anc" % s
        os.remove(anc)
        os.remove(os.path.join(d, "out_%d.mut" % s))


def test_paint_stage_writes_the_references_files_at_full_length(pinned):
    z, work, (N, L, W) = pinned
    run(work, "--mode", "Paint")
    pdir = os.path.join(work, "out", "chunk_0", "paint")
    bad = [w for w in range(W) if os.path.getsize(os.path.join(pdir, "relate_%d.bin" % w)) != int(z["paint_size"][w]) or
           not np.array_equal(md5_file(os.path.join(pdir, "relate_%d.bin" % w)), z["paint_md5"][w])]
    assert not bad, "paint files of windows %s differ from the reference's" % bad


def test_build_topology_of_three_sections_from_those_files(pinned):
    z, work, (N, L, W) = pinned
    for s in (int(x) for x in z["sections"]):
        run(work, "--mode", "BuildTopology", "--first_section", str(s), "--last_section", str(s))
    check_sections(z, work)


def test_fused_stage_of_the_whole_chunk_writes_the_same_sections(pinned):
    """Paint + BuildTopology of ALL sections in one call, no paint files, trees on the device"""
    z, work, (N, L, W) = pinned
    run(work, "--mode", "PaintBuildTopology", "--first_section", "0", "--last_section", str(W - 1))
    check_sections(z, work)
