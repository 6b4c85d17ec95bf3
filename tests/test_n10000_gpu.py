"""BASELINE.json config #5's N against the reference binary: N = 10,000 haplotypes -- two wavefronts of S = 80
registers per target in K1 / K2 (N > 5120), K3's gather from that layout, the device tree builder's state in global
memory (N > ~5200) -- on a chunk short enough for the single-threaded reference (L = 600, 3 windows; about ten minutes
of it).  tests/golden/n10000.npz (tools/make_golden.py n10000) holds the reference's md5 of every paint file, of section
0's .anc / .mut, the .mut in full and the parent arrays of its trees (md5 each, three in full); the inputs are
regenerated from the seed and md5-checked.

Held to those bytes: the drop-in CLI (Paint; BuildTopology with the host and the device builder; the fused stage) and
the route of config #5 itself -- relate_amd.dist.run_chunk_by_targets with the chunk cut into 2 and 3 target ranges,
the "ranks" being threads on this one GPU (ThreadFabric; the exchange protocol over gloo: tests/test_target_shard_cpu.py)."""
import os
import threading

import numpy as np
import pytest

import bigtile
from relate_amd import api, dist as rdist

pytestmark = pytest.mark.gpu

GOLD = os.path.join(bigtile.ROOT, "tests", "golden", "n10000.npz")


@pytest.fixture(scope="module")
def chunk(tmp_path_factory):
    work = str(tmp_path_factory.mktemp("n10000"))
    z, W = bigtile.make_chunk_dir(GOLD, work)
    return z, work, W


def test_paint_files_at_config5_N(chunk):
    z, work, W = chunk
    bigtile.run_cli("Paint", work, "host")
    ctx = api.Context()
    ctx.load_chunk(os.path.join(work, "out"), 0)
    assert (ctx.N, ctx.tile, ctx.waves) == (10000, 80, 2)
    ctx.close()
    for w in range(W):
        b = open(os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % w), "rb").read()
        if w == 0:
            head = z["head/paint/relate_0.bin"].tobytes()
            assert b[:len(head)] == head, "window 0's paint file differs within its first %d bytes" % len(head)
        assert np.array_equal(bigtile.md5(b), z["md5/paint/relate_%d.bin" % w]), "paint file of window %d" % w


@pytest.mark.parametrize("builder", ["host", "gpu"])
def test_section_0_trees_at_config5_N(chunk, builder):
    z, work, W = chunk
    if not os.path.exists(os.path.join(work, "out", "chunk_0", "paint", "relate_0.bin")):
        bigtile.run_cli("Paint", work, "host")
    err = bigtile.run_cli("BuildTopology", work, builder)
    if builder == "gpu":
        assert "(0 trees on the GPU" not in err, err[-800:]
    bigtile.check_section_0(z, os.path.join(work, "out"))


def test_fused_stage_at_config5_N(chunk, tmp_path):
    z, work, W = chunk
    bigtile.link_inputs(os.path.join(work, "out"), str(tmp_path / "out"))
    bigtile.run_cli("PaintBuildTopology", str(tmp_path), "gpu")
    assert not os.path.exists(str(tmp_path / "out" / "chunk_0" / "paint"))
    bigtile.check_section_0(z, str(tmp_path / "out"))


@pytest.mark.parametrize("parts,build_on_gpu", [(2, True), (3, False)])
def test_chunk_sharded_by_target_reproduces_the_reference(chunk, tmp_path, parts, build_on_gpu):
    """config #5's route on one GPU: `parts` ranks (threads), each painting its own range of targets (uneven for 3:
    3334 + 3333 + 3333), section 0's trees built by its owner from matrices assembled out of every rank's rows"""
    z, work, W = chunk
    out = str(tmp_path / "out")
    bigtile.link_inputs(os.path.join(work, "out"), out)
    hub = rdist.ThreadFabric.Hub(parts)
    res, errs = [None] * parts, [None] * parts

    def body(r):
        try:
            res[r] = rdist.run_chunk_by_targets(out, 0, device=0, sections=[0], in_flight=1, build_on_gpu=build_on_gpu,
                                                fabric=rdist.ThreadFabric(hub, r, device=0))
        except BaseException as e:
            errs[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs == [None] * parts, errs
    assert res[0] == {0: len(z["tree_pos"])} and all(x == {} for x in res[1:])
    bigtile.check_section_0(z, out)
