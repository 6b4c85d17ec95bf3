"""The exchange protocol of the target-sharded chunk (BASELINE.json config #5, relate_amd.dist.run_chunk_by_targets)
without a GPU: stand-in shards whose "distance rows" are a known function of (section, SNP, target, donor), stand-in
section owners that ask for a known sequence of SNPs and check every assembled matrix -- through threads of one process
(ThreadFabric: what the one-GPU test uses) and through a 2-rank gloo job (TorchFabric: what RCCL runs), with N not a
multiple of the number of ranks, more sections than ranks, several sections in flight per rank, and a failing rank."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relate_amd import dist as rdist  # noqa: E402


def expected_matrix(N, section, snp):
    k = np.arange(N, dtype=np.float32)[:, None]
    n = np.arange(N, dtype=np.float32)[None, :]
    return (1000.0 * section + snp + 0.25 * k + 0.001 * n).astype(np.float32)


class FakeShard:
    """api.Shard's interface; rows() insists on the call order the real one needs (SNPs never decrease within a
    section, no rows after a release)"""

    def __init__(self, N, W, rank, world, fail_at=None):
        self.N, self.W = N, W
        self.k_begin, self.k_end = rdist.target_range(rank, world, N)
        self.last = {}
        self.released = []
        self.rows_calls = 0
        self.fail_at = fail_at
        self.lock = threading.Lock()

    def section_bounds(self, section):
        return 100 * section, 100 * section + 99

    def set_window_rows(self, rows):
        self.window_rows = rows

    def expect_builders(self, n):
        pass

    def rows(self, section, snp, ptr):
        with self.lock:
            assert section not in self.released
            assert snp >= self.last.get(section, -1), "SNPs of a section must not decrease"
            self.last[section] = snp
            self.rows_calls += 1
            if self.fail_at is not None and self.rows_calls == self.fail_at:
                raise RuntimeError("injected failure")
        n = self.k_end - self.k_begin
        out = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_float)), shape=(n, self.N))
        out[:] = expected_matrix(self.N, section, snp)[self.k_begin:self.k_end]

    def release_section(self, section):
        with self.lock:
            self.released.append(section)

    def copy_on_device(self, dst, src, nbytes):
        C.memmove(dst, src, nbytes)

    def build_section(self, section, matrix, matrix_dev=None, build_device=None, no_consistency=False, fb=0):
        a, b = self.section_bounds(section)
        d = np.full((self.N, self.N), -1.0, dtype=np.float32)
        trees = 0
        for snp in [a, a, a + 3 + section, a + 50, b]:  # (the same SNP twice: a tree handed from the device to the host)
            d[:] = -1.0
            assert matrix(snp, d.ctypes.data) in (None, 0)
            assert np.array_equal(d, expected_matrix(self.N, section, snp)), (section, snp)
            trees += 1
        return trees


def run_rank(fab, N, W, fail_at=None, in_flight=2, sections=None):
    sh = FakeShard(N, W, fab.rank, fab.world, fail_at)
    res = rdist.run_chunk_by_targets("unused", 0, fabric=fab, shard=sh, in_flight=in_flight, sections=sections,
                                     idle_sleep=0.0)
    return res, sh


@pytest.mark.parametrize("world,N,W,in_flight", [(3, 10, 7, 2), (2, 9, 5, 3), (1, 5, 3, 1), (4, 6, 2, 2)])
def test_protocol_over_threads(world, N, W, in_flight):
    hub = rdist.ThreadFabric.Hub(world)
    out = [None] * world

    def body(r):
        out[r] = run_rank(rdist.ThreadFabric(hub, r), N, W, in_flight=in_flight)

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    owned = {}
    for r in range(world):
        res, sh = out[r]
        assert sorted(res) == rdist.deal_sections(list(range(W)), r, world)
        assert all(v == 5 for v in res.values())
        assert sorted(sh.released) == list(range(W))        # every rank closed every section's window
        assert sh.rows_calls == 5 * W                       # ... and served its rows of every matrix of the job
        owned.update(res)
    assert sorted(owned) == list(range(W))


@pytest.mark.parametrize("bad_rank,fail_at", [(1, 7), (0, 1), (2, 4), (1, 12), (0, 9), (2, 17)])
def test_a_failing_rank_stops_every_rank(bad_rank, fail_at):
    """... and no owner ever gets a matrix with the failing rank's stale rows in it (the blocks carry a flag row):
    FakeShard.build_section would raise an AssertionError, not the job's RuntimeError"""
    world, N, W = 3, 10, 6
    hub = rdist.ThreadFabric.Hub(world)
    errors = [None] * world

    def body(r):
        try:
            run_rank(rdist.ThreadFabric(hub, r), N, W, fail_at=fail_at if r == bad_rank else None)
        except BaseException as e:
            errors[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert all(isinstance(e, RuntimeError) for e in errors), errors
    assert "injected" in str(errors[bad_rank])


def _gloo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, sh = run_rank(rdist.TorchFabric(), 9, 5, in_flight=2)
    q.put((rank, sorted(res), sorted(sh.released), sh.rows_calls))
    dist.barrier()
    dist.destroy_process_group()


def test_protocol_over_gloo_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, [0, 2, 4], [0, 1, 2, 3, 4], 25)
    assert res[1] == (1, [1, 3], [0, 1, 2, 3, 4], 25)
