"""One chunk sharded by target haplotype (BASELINE.json config #5): contexts that
each paint a range of targets reproduce, row for row and bit for bit, what a
context with all targets computes -- stepping stones, posteriors, distance
rows -- including through the device-buffer path that feeds the all-gather."""
import numpy as np
import pytest

import rlutil
from relate_amd import api, dist as rdist

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,L,budget,parts", [(130, 900, 200000, 2), (70, 700, 40000, 3), (5300, 90, None, 3)])
@pytest.mark.parametrize("mode", ["exact", "lanes"])
def test_target_ranges_reproduce_the_full_chunk(N, L, budget, parts, mode):
    sm = api.RL_SUM_EXACT if mode == "exact" else api.RL_SUM_LANES
    if budget is None:  # (N > 5120: two wavefronts per target)
        from test_edge_gpu import random_chunk
        ch = random_chunk(N, L, 0.13, seed=N, wb=[0, 30, 60, L])
    else:
        ch = rlutil.synth_chunk(N, L, seed=4, budget=budget)
    full = api.Context()
    full.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
    full.paint(sm)
    W = ch.W
    w = W // 2
    snp = int(ch.wb[w]) + 3
    fst = full.stones(w)
    fwin = full.open_window(w, None, int(ch.wb[w]), sm)
    for s in range(int(ch.wb[w]) + 1, snp + 1):
        fwin.advance(s)
    fd = fwin.matrix(snp)
    total = 0
    rows = []
    for r in range(parts):
        k0, k1 = rdist.target_range(r, parts, N)
        ctx = api.Context()
        ctx.set_chunk(ch.seq, ch.r, ch.rpos, ch.wb)
        ctx.set_target_range(k0, k1)
        assert ctx.target_range() == (k0, k1)
        total += ctx.total_sites()
        ctx.paint(sm)
        st = ctx.stones(w)
        for key in ("alpha", "beta", "ls_alpha", "ls_beta", "bsnp_begin", "bsnp_end"):
            assert st[key].shape[0] == k1 - k0
            assert np.array_equal(st[key].view(np.uint32) if st[key].dtype == np.float32 else st[key],
                                  fst[key][k0:k1].view(np.uint32) if fst[key].dtype == np.float32 else fst[key][k0:k1]), key
        win = ctx.open_window(w, None, int(ch.wb[w]), sm)
        for s in range(int(ch.wb[w]) + 1, snp + 1):
            win.advance(s)
        d = win.matrix(snp)
        assert d.shape == (k1 - k0, N)
        assert np.array_equal(d.view(np.uint32), fd[k0:k1].view(np.uint32))
        top, ls = win.topology(k0)
        ftop, fls = fwin.topology(k0)
        assert np.array_equal(top.view(np.uint32), ftop.view(np.uint32)) and np.array_equal(ls, fls)
        with pytest.raises(api.RelateError):
            ctx.write_paint_files("/tmp/never_written")
        # the device-buffer path (send buffer of the all-gather)
        import torch
        buf = torch.empty((k1 - k0, N), dtype=torch.float32, device="cuda")
        win.matrix_rows_into(snp, buf.data_ptr())
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy())
        win.close()
        ctx.close()
    assert total == full.total_sites()
    assert np.array_equal(np.concatenate(rows, 0).view(np.uint32), fd.view(np.uint32))
    fwin.close()
    full.close()


@pytest.mark.parametrize("name,parts,build_on_gpu,in_flight,from_files", [
    ("synth70", 2, True, 2, False), ("synth70", 3, False, 3, False), ("synth24", 4, True, 1, True),
    ("synth40_noisy", 3, True, 4, False), ("synth70", 8, True, 2, False)])
def test_sharded_route_writes_the_reference_files(tmp_path, name, parts, build_on_gpu, in_flight, from_files):
    """relate_amd.dist.run_chunk_by_targets (config #5's route) on golden chunks: `parts` ranks (threads on this GPU)
    paint their target ranges, the sections are dealt to them as owners, every matrix is assembled from all ranks' rows
    -- every section's .anc / .mut byte-identical to the reference binary's (from_files: the shards read their
    targets' records from the reference's paint files instead of painting)"""
    import threading
    from golden_util import Fixture
    out = tmp_path / "out"
    out.mkdir()
    fx = Fixture(name, out)
    if from_files:
        fx.write_paint_files(str(out / "chunk_0" / "paint"))
    hub = rdist.ThreadFabric.Hub(parts)
    res, errs = [None] * parts, [None] * parts

    def body(r):
        try:
            res[r] = rdist.run_chunk_by_targets(str(out), 0, device=0, in_flight=in_flight, build_on_gpu=build_on_gpu,
                                                from_paint_files=from_files, fabric=rdist.ThreadFabric(hub, r, device=0))
        except BaseException as e:
            errs[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs == [None] * parts, errs
    assert sorted(s for r in res for s in r) == list(range(fx.W))
    for w in range(fx.W):
        assert open(out / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(out / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["anc/%d" % w].tobytes(), w
