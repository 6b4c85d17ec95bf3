"""The headline tile against the reference binary: N = 5000 haplotypes (BASELINE.json config #3's N: one wavefront
of S = 80 registers per target), a chunk short enough for the single-threaded reference to finish in minutes
(L = 1200, 2 windows).  tests/golden/n5000.npz (tools/make_golden.py n5000) holds the reference's md5 of every
output file of Paint (whole chunk) and BuildTopology (section 0), the head of window 0's paint file, the .mut file
and the parent arrays (md5 per tree, three in full); the inputs are regenerated from the seed and checked against
their md5s.  Through the drop-in CLI, with the trees built on the host and on the GPU."""
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


def md5(b):
    return np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def painted(tmp_path_factory):
    z = np.load(GOLD)
    N, L, W, seed = [int(x) for x in z["meta"]]
    mem = float(z["mem"][0])
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget),
                                wb.ctypes.data_as(C.c_void_p), 499) == W
    assert np.array_equal(wb[:W + 1], z["wb"])
    work = tmp_path_factory.mktemp("n5000")
    d = os.path.join(str(work), "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p),
                                    bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                    rpos.ctypes.data_as(C.c_void_p), wb.ctypes.data_as(C.c_void_p), W) == 0
    for k in z.files:  # the chunk files are the ones the reference was given
        if k.startswith("in_md5/"):
            assert np.array_equal(md5(open(os.path.join(d, k[7:]), "rb").read()), z[k]), k
    p = subprocess.run([CLI, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=str(work),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return z, str(work), W


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
    env = dict(os.environ)
    env["RELATE_AMD_GPU_BUILD"] = "1" if builder == "gpu" else "0"
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", "0", "-o", "out"], cwd=work, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()
    anc = os.path.join(work, "out", "chunk_0", "out_0.anc")
    mut = open(os.path.join(work, "out", "chunk_0", "out_0.mut"), "rb").read()
    _, trees = rlutil.parse_anc(anc)
    assert [t[0] for t in trees] == list(z["tree_pos"]), "tree positions"
    for t, (tr, want) in enumerate(zip(trees, z["tree_parent_md5"])):
        if "tree_parent/%d" % t in z.files:
            assert np.array_equal(tr[1], z["tree_parent/%d" % t]), "parent array of tree %d" % t
        assert np.array_equal(md5(tr[1].astype("<i4").tobytes()), want), "parent array of tree %d" % t
    assert mut == z["mut/0"].tobytes()
    assert np.array_equal(md5(mut), z["md5/out_0.mut"])
    assert np.array_equal(md5(open(anc, "rb").read()), z["md5/out_0.anc"])
