"""world_size-2 gloo test of the chunk-sharded multi-GPU path (CPU only).

Each rank takes its share of the chunks and "paints" them -- with the oracle
standing in as the per-chunk worker, since there is no GPU here -- and the
job statistics are reduced exactly as bench.py does with RCCL."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rlutil
    from relate_amd import dist as rdist
    o = rlutil.oracle()
    chunks = list(range(5))
    mine = rdist.shard(chunks, rank, world)
    sites = 0
    for c in mine:
        ch = rlutil.synth_chunk(16, 300, seed=100 + c, budget=None)
        d = ch.ro()
        s = C.c_longlong(0)
        assert o.ro_paint_chunk(C.byref(d), ch.wb.ctypes.data_as(C.c_void_p), ch.W, None, 1, 0, None,
                                C.byref(s)) == 0
        sites += s.value
    total, tmax = rdist.job_stats(sites, 1.0 + rank)
    q.put((rank, mine, sites, total, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_chunk_sharding_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, s0, t0, x0), (r1, m1, s1, t1, x1) = res
    assert m0 == [0, 2, 4] and m1 == [1, 3]          # disjoint cover of the chunks
    assert t0 == t1 == s0 + s1                        # whole-job units
    assert x0 == x1 == 2.0                            # max over ranks


def test_shard_is_a_partition():
    from relate_amd import dist as rdist
    for world in (1, 2, 3, 8):
        parts = [rdist.shard(list(range(11)), r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(11))


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    from relate_amd import dist as rdist
    N = 11  # not a multiple of the world size: uneven blocks
    full = torch.arange(N * N, dtype=torch.float32).reshape(N, N) * 0.5
    k0, k1 = rdist.target_range(rank, world, N)
    got = rdist.all_gather_rows(full[k0:k1].clone(), N)  # each rank contributes only its own rows
    q.put((rank, k0, k1, bool(torch.equal(got, full))))
    dist.barrier()
    dist.destroy_process_group()


def test_target_sharding_all_gather_three_ranks():
    """the single-chunk route (config #5): ranks own target ranges and all-gather their distance rows"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 8), (8, 11)]
    assert all(r[3] for r in res)


def test_target_range_is_a_partition():
    from relate_amd import dist as rdist
    for world in (1, 2, 3, 8):
        for N in (8, 11, 5000):
            rs = [rdist.target_range(r, world, N) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == N
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def _chunk_worker(rank, world, port, q, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from relate_amd import dist as rdist

    class Stages:  # records what each rank is asked to do (the real stages need a GPU)
        calls = []

        def stage_paint(self, out, c, painting=None, device=0):
            self.calls.append(("paint", c))
            open(os.path.join(out, "painted"), "w").close()

        def num_sections(self, out, c):
            return 7

        def stage_build_topology(self, out, c, first, last, painting=None, device=0):
            assert os.path.exists(os.path.join(out, "painted"))  # only after rank 0's paint stage
            self.calls.append(("build", first, last))
            open(os.path.join(out, "built_%d" % dist.get_rank()), "w").close()

        def stage_find_equivalent_branches(self, out, c):
            assert all(os.path.exists(os.path.join(out, "built_%d" % r)) for r in range(dist.get_world_size()))
            self.calls.append(("feb", c))

    st = Stages()
    rng = rdist.run_chunk(out_dir, 0, stages=st)
    q.put((rank, rng, st.calls))
    dist.destroy_process_group()


def test_chunk_pipeline_shards_sections_three_ranks(tmp_path):
    """Paint on rank 0, BuildTopology sections split over the ranks, FindEquivalentBranches on rank 0, barriers between"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [(0, 2), (3, 4), (5, 6)]
    assert res[0][2] == [("paint", 0), ("build", 0, 2), ("feb", 0)]
    assert res[1][2] == [("build", 3, 4)] and res[2][2] == [("build", 5, 6)]


def _chunks_worker(rank, world, port, q, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from relate_amd import dist as rdist

    class Stages:  # records what each rank is asked to do (the real stages need a GPU)
        calls = []

        def stage_paint(self, out, c, painting=None, device=0):
            assert device == dist.get_rank()  # LOCAL_RANK picks the GPU
            self.calls.append(("paint", c))

        def num_sections(self, out, c):
            return 3 + c

        def stage_build_topology(self, out, c, first, last, painting=None, device=0):
            assert self.calls[-1] == ("paint", c)
            self.calls.append(("build", c, first, last))

        def stage_paint_build_topology(self, out, c, first, last, painting=None, device=0):
            assert device == dist.get_rank()
            self.calls.append(("fused", c, first, last))

        def stage_find_equivalent_branches(self, out, c):
            self.calls.append(("feb", c))

    st = Stages()
    mine = rdist.run_chunks(out_dir, stages=st)  # the default: one fused stage per chunk, no paint files
    fused_calls = list(st.calls)
    del st.calls[:]
    assert rdist.run_chunks(out_dir, stages=st, paint_files=True) == mine
    q.put((rank, mine, st.calls, fused_calls))
    dist.destroy_process_group()


def test_many_chunks_dealt_to_ranks(tmp_path):
    """config #4's route: the chunks of parameters.bin dealt round-robin, each rank runs its chunks start to end"""
    import struct
    from relate_amd import dist as rdist
    C = 7
    with open(tmp_path / "parameters.bin", "wb") as f:
        f.write(struct.pack("<iii", 2000, 5000000, C) + struct.pack("<d", 1.0))
        f.write(struct.pack("<%di" % C, *[max(0, 100000 * c - 20000) for c in range(C)]))
        f.write(struct.pack("<%di" % C, *[100000 * (c + 1) for c in range(C)]))
    par = rdist.read_parameters(str(tmp_path))
    assert (par["N"], par["L"], par["num_chunks"], par["start"][1], par["end"][-1]) == (2000, 5000000, C, 80000, 700000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_chunks_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0, 3, 6], [1, 4], [2, 5]]              # a disjoint cover of the chunks
    assert sorted(sum((r[1] for r in res), [])) == list(range(C))
    for _, mine, calls, fused in res:
        assert calls == sum(([("paint", c), ("build", c, 0, 2 + c), ("feb", c)] for c in mine), [])
        assert fused == sum(([("fused", c, 0, 2 + c), ("feb", c)] for c in mine), [])


def test_dist_command_line_parses():
    """python -m relate_amd.dist OUT_DIR [--painting theta,rho] [--chunks a,b]: the many-chunks runner's entry"""
    import pytest as _pytest
    from relate_amd import dist as rdist
    with _pytest.raises(SystemExit) as e:
        rdist.main(["--help"])
    assert e.value.code == 0
    with _pytest.raises(SystemExit):
        rdist.main([])  # the output directory is required


def test_launcher_really_starts_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m relate_amd.dist OUT` as a user types it (gloo: no GPU
    here), the stages replaced by recorders (--stages): both ranks come up, rendezvous, are dealt their chunks, run
    them start to end on the device LOCAL_RANK names and leave together"""
    import socket
    import struct
    import subprocess
    out = tmp_path / "job"
    out.mkdir()
    C_ = 5
    with open(out / "parameters.bin", "wb") as f:  # data.cpp:365-375
        f.write(struct.pack("<iii", 40, 1000, C_) + struct.pack("<d", 5.0) + struct.pack("<%di" % (2 * C_), *range(2 * C_)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "relate_amd.dist", str(out),
                        "--stages", "stubs.stub_stages"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env,
                       timeout=300)
    text = p.stdout.decode()
    assert p.returncode == 0, text[-2000:]
    assert "rank 0 ran chunks [0, 2, 4]" in text and "rank 1 ran chunks [1, 3]" in text, text[-2000:]
    calls0 = open(out / "calls_rank0.txt").read().split("\n")
    calls1 = open(out / "calls_rank1.txt").read().split("\n")
    assert calls0[:2] == ["0 fused 0 0-2 dev0", "0 feb 0"] and "0 fused 4 0-6 dev0" in calls0
    assert calls1[:2] == ["1 fused 1 0-3 dev1", "1 feb 1"] and "1 fused 3 0-5 dev1" in calls1


def test_launcher_by_targets_three_ranks(tmp_path):
    """the route of config #5 as a user starts it: `python -m torch.distributed.run --nproc-per-node 3 -m relate_amd.dist
    OUT --by-targets` (gloo; stand-in shards): three real processes exchange request tables and row blocks (N = 10 over
    3 ranks: uneven), every section is built by its owner from matrices assembled out of all ranks' rows and released
    everywhere, rank 0 runs the host-only stage after the barrier"""
    import socket
    import struct
    import subprocess
    out = tmp_path / "job"
    out.mkdir()
    with open(out / "parameters.bin", "wb") as f:
        f.write(struct.pack("<iii", 10, 1000, 1) + struct.pack("<d", 5.0) + struct.pack("<2i", 0, 1000))
    with open(out / "parameters_c0.bin", "wb") as f:
        f.write(struct.pack("<iii", 10, 1000, 4))  # (stub: N = 10; the stub's num_sections gives 3 + chunk)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "relate_amd.dist", str(out),
                        "--by-targets", "--in-flight", "2", "--stages", "stubs.stub_stages"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, env=env, timeout=300)
    text = p.stdout.decode()
    assert p.returncode == 0, text[-2500:]
    for r, owned in ((0, [0]), (1, [1]), (2, [2])):
        assert "rank %d of 3: chunk 0 by targets, owned sections %s" % (r, owned) in text, text[-2500:]
        calls = open(out / ("calls_rank%d.txt" % r)).read().split("\n")
        assert "%d built 0 %d" % (r, owned[0]) in calls
        assert sorted(c for c in calls if " release " in c) == ["%d release 0 %d" % (r, s) for s in range(3)]
    assert "0 feb 0" in open(out / "calls_rank0.txt").read()


def test_bench_self_launch_command():
    """`python bench.py --gpus 4` without a launcher starts its ranks itself, as a child: the command it would run
    (--print-launch; the GPUs to run it on are not here)"""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--print-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    cmd = json.loads(p.stdout.decode().strip().split("\n")[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
