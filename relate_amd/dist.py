"""Multi-GPU partitioning of the Paint -> BuildTopology path.

The path shards by genomic chunk (one process per GPU, no data-path
collective): chunks are independent after MakeChunks, exactly as the
reference's cluster scripts run them (scripts/RelateParallel/RelateParallel.sh:216,
scripts/RelateSGE/RelateSGE.sh:324-401).  Within a chunk, BuildTopology shards
by section (window) the same way (RelateParallel.sh:231-257).  The only
collectives on that route are the bookkeeping ones below (job statistics).

A single chunk too large for one GPU (BASELINE.json config #5) shards by TARGET
haplotype instead: every rank holds the bit panel, paints / re-paints its own
contiguous range of targets (rl_set_target_range) and owns those rows of every
distance matrix; one all-gather (RCCL over xGMI) assembles the N x N matrix for
the tree builder (target_range, all_gather_rows)."""
import os
import struct

import torch
import torch.distributed as dist


def shard(items, rank, world):
    """round-robin assignment of chunk (or section) indices to ranks"""
    return [x for i, x in enumerate(items) if i % world == rank]


def job_stats(units, seconds, device=None):
    """whole-job aggregate: (sum of units over ranks, max of seconds over ranks)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(units), float(seconds)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    u = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(u.item()), float(t.item())


def target_range(rank, world, N):
    """contiguous, balanced range [k_begin, k_end) of targets for a rank (the first N % world ranks get one more)"""
    q, r = divmod(N, world)
    k0 = rank * q + min(rank, r)
    return k0, k0 + q + (1 if rank < r else 0)


def all_gather_rows(rows, N):
    """rows: this rank's (k_end-k_begin) x N block of a distance matrix, a tensor on the backend's device.
    Returns the N x N matrix (every rank).  Row counts may differ by one between ranks, so blocks are
    padded to the largest and the padding dropped after the gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    maxrows = -(-N // world)
    send = rows
    if rows.shape[0] < maxrows:
        send = torch.zeros((maxrows, N), dtype=rows.dtype, device=rows.device)
        send[: rows.shape[0]] = rows
    out = torch.empty((world * maxrows, N), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, send.contiguous())
    blocks = []
    for r in range(world):
        k0, k1 = target_range(r, world, N)
        blocks.append(out[r * maxrows: r * maxrows + (k1 - k0)])
    return torch.cat(blocks, 0)


def section_ranges(num_sections, rank, world):
    """contiguous, balanced [first, last] range of BuildTopology sections for a rank, or None (rl_stage_build_topology
    takes a range and runs its sections on host threads; sections are independent, RelateParallel.sh:231-257)"""
    k0, k1 = target_range(rank, world, num_sections)
    return (k0, k1 - 1) if k1 > k0 else None


def run_chunk(out_dir, chunk_index=0, painting=None, device=None, stages=None):
    """Paint -> BuildTopology -> FindEquivalentBranches of one chunk on all ranks of the job (one process per GPU):
    rank 0 paints and writes the paint files, every rank builds its share of the sections from them, rank 0
    runs the (host-only) branch association.  The only synchronisation is a barrier between the stages -- the files
    are the interface, exactly as between the reference's cluster jobs.  `stages` = object with stage_paint,
    stage_build_topology, stage_find_equivalent_branches, num_sections (default: relate_amd.api)."""
    if stages is None:
        from relate_amd import api as stages
    live = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if live else 0
    world = dist.get_world_size() if live else 1
    dev = device if device is not None else local_device()
    if rank == 0:
        stages.stage_paint(out_dir, chunk_index, painting=painting, device=dev)
    if live:
        dist.barrier()
    rng = section_ranges(stages.num_sections(out_dir, chunk_index), rank, world)
    if rng is not None:
        stages.stage_build_topology(out_dir, chunk_index, rng[0], rng[1], painting=painting, device=dev)
    if live:
        dist.barrier()
    if rank == 0:
        stages.stage_find_equivalent_branches(out_dir, chunk_index)
    if live:
        dist.barrier()
    return rng


def local_device():
    """the GPU of this process: LOCAL_RANK under torch.distributed.run (one process per GPU of the node), else 0"""
    return int(os.environ.get("LOCAL_RANK", "0"))


def read_parameters(out_dir):
    """parameters.bin of MakeChunks (data.cpp:365-375): int N, L, num_chunks; double memory; int start[], end[]
    -> dict(N, L, num_chunks, memory_gb, start, end)"""
    buf = open(os.path.join(out_dir, "parameters.bin"), "rb").read()
    if len(buf) < 20:
        raise ValueError("%s/parameters.bin is malformed" % out_dir)
    N, L, C = struct.unpack_from("<iii", buf, 0)
    mem, = struct.unpack_from("<d", buf, 12)
    if C < 1 or len(buf) < 20 + 8 * C:
        raise ValueError("%s/parameters.bin is malformed" % out_dir)
    start = list(struct.unpack_from("<%di" % C, buf, 20))
    end = list(struct.unpack_from("<%di" % C, buf, 20 + 4 * C))
    return dict(N=N, L=L, num_chunks=C, memory_gb=mem, start=start, end=end)


def run_chunks(out_dir, painting=None, device=None, stages=None, chunks=None, paint_files=False, timings=None):
    """The many-chunks route (BASELINE.json config #4; scripts/RelateParallel/RelateParallel.sh:216-262): the chunks of
    a MakeChunks directory are dealt round-robin to the ranks of the job (one process per GPU) and every rank runs
    its chunks start to end -- Paint + BuildTopology of all sections in ONE stage call with the stepping stones kept
    in HBM (rl_stage_paint_build_topology; paint_files=True takes the reference's two stages with the paint files in
    between: 12.8 GB per C4-sized chunk, eight ranks on one filesystem), then FindEquivalentBranches -- with NO
    data-path collective and no barrier: chunks share nothing but the input directory, a rank is done when its
    chunks are.  Returns the chunk indices this rank ran.  `chunks`: a subset to run (default: all of parameters.bin).
    `timings`: a list that receives (chunk, stage name, seconds) per stage call (tools/c4_job_one_gpu.py)."""
    import time
    if stages is None:
        from relate_amd import api as stages
    live = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if live else 0
    world = dist.get_world_size() if live else 1
    dev = device if device is not None else local_device()
    todo = list(range(read_parameters(out_dir)["num_chunks"])) if chunks is None else list(chunks)
    mine = shard(todo, rank, world)
    def timed(c, name, fn, *a, **kw):
        t0 = time.time()
        fn(*a, **kw)
        if timings is not None:
            timings.append((c, name, time.time() - t0))

    for c in mine:
        last = stages.num_sections(out_dir, c) - 1
        if paint_files:
            timed(c, "paint", stages.stage_paint, out_dir, c, painting=painting, device=dev)
            timed(c, "build_topology", stages.stage_build_topology, out_dir, c, 0, last, painting=painting, device=dev)
        elif getattr(stages, "FUSED_FEB", False) and last > 0:
            # (the GPU library: FindEquivalentBranches runs on the trees while they are in memory, every .anc is
            #  written once -- same bytes as the two stages one after the other)
            timed(c, "paint_build_topology_feb", stages.stage_paint_build_topology, out_dir, c, 0, last,
                  painting=painting, device=dev, find_equivalent_branches=True)
            continue
        else:
            timed(c, "paint_build_topology", stages.stage_paint_build_topology, out_dir, c, 0, last, painting=painting,
                  device=dev)
        timed(c, "find_equivalent_branches", stages.stage_find_equivalent_branches, out_dir, c)
    return mine


# ---- one chunk sharded by target haplotype (BASELINE.json config #5) -------------------------------------------
#
# Units (include/relate_amd.h "one chunk sharded by target haplotype"; what they mirror in the reference:
# src/anc_builder.cpp:49-106 -- a section needs the stones of EVERY target -- and RelateParallel.sh:231-257 -- sections
# are the independent jobs): every rank is an api.Shard for its target_range(); the sections are dealt to the ranks
# as OWNERS; an owner runs the section's tree-sequence loop and, for every tree, asks all ranks for their rows of the
# distance matrix at one SNP.  Collectives need one order on all ranks, so requests are exchanged in TICKS:
#
#   tick:  every rank publishes its table of <= in_flight requests (section, snp, kind)        -- all_gather, 24 B each
#          for every request of the tick, in (rank, slot) order:
#              every rank: shard.rows(section, snp) -> its (maxrows x N) send block             -- K2 (first time) + K3
#              all ranks:  all_gather of the blocks (N^2 floats in all; RCCL over xGMI)         -- THE exchange
#              the owner:  the blocks, pads dropped, into the N x N buffer its builder waits for
#          kind RELEASE: every rank closes its window of that section
#   the job ends in the tick in which every rank reports that it owns nothing any more.
#
# Bytes per tree at N = 10,000 on 8 GPUs: each rank sends 1250 x 10,000 x 4 B = 50 MB and receives 350 MB (2.3 ms at
# the ~153 GB/s of one xGMI link in a ring; a tree's build is ~0.5 s).  Nothing else of the chunk moves: stones
# (36 GB per rank), posterior rows (3 GB per window and rank) and cursors stay where they were computed.
REQ_NONE, REQ_MATRIX, REQ_RELEASE = 0, 1, 2


class TorchFabric:
    """the job's ranks = the ranks of torch.distributed (RCCL on GPUs, gloo on the CPU)"""

    def __init__(self, device=None):
        self.live = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.live else 0
        self.world = dist.get_world_size() if self.live else 1
        on_gpu = self.live and dist.get_backend() == "nccl"
        self.device = torch.device("cuda", device if device is not None else local_device()) if (
            on_gpu or (device is not None and torch.cuda.is_available())) else torch.device("cpu")

    def buffer(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float32, device=self.device)

    def all_gather_table(self, table):
        t = torch.as_tensor(table, dtype=torch.int64).to(self.device)
        if not self.live or self.world == 1:
            return t.cpu().numpy()[None]
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out.cpu().numpy().reshape((self.world,) + tuple(t.shape))

    def all_gather_rows(self, send, recv):
        self.all_gather_rows_many([send], [recv])

    def all_gather_rows_many(self, sends, recvs):
        """the blocks of SEVERAL matrices: the collectives queue up behind one another, ONE wait for all of them"""
        for send, recv in zip(sends, recvs):
            if not self.live or self.world == 1:
                recv.copy_(send)
            else:
                dist.all_gather_into_tensor(recv, send)
        if recvs and recvs[0].is_cuda:
            # (the stream the collectives were ordered on, NOT the device: a device-wide wait also waits for the tree
            #  builder's resident workers, i.e. for every other section's tree -- 0.6 s per matrix at N = 10,000)
            torch.cuda.current_stream(recvs[0].device).synchronize()


class ThreadFabric:
    """the job's ranks = threads of this process (several target ranges on ONE GPU, or none: tests).  Same protocol,
    the exchanges through shared lists and a barrier."""

    class Hub:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, hub, rank, device=None):
        self.hub, self.rank, self.world = hub, rank, hub.world
        self.device = torch.device("cuda", device) if device is not None else torch.device("cpu")

    def buffer(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float32, device=self.device)

    def _exchange(self, item):
        self.hub.slots[self.rank] = item
        self.hub.barrier.wait()
        items = list(self.hub.slots)
        self.hub.barrier.wait()
        return items

    def all_gather_table(self, table):
        import numpy as np
        return np.stack(self._exchange(np.array(table, dtype=np.int64)))

    def all_gather_rows(self, send, recv):
        self.all_gather_rows_many([send], [recv])

    def all_gather_rows_many(self, sends, recvs):
        """the blocks of several matrices with ONE exchange, one wait and one closing barrier"""
        blocks = self._exchange(list(sends))
        for i, recv in enumerate(recvs):
            n = sends[i].shape[0]
            for r, b in enumerate(blocks):
                recv[r * n:(r + 1) * n].copy_(b[i])
        if recvs and recvs[0].is_cuda:
            # (the stream the copies were ordered on, NOT the device: a device-wide wait also waits for the tree
            #  builder's resident workers, i.e. for every other section's tree -- 0.6 s per matrix at N = 10,000)
            torch.cuda.current_stream(recvs[0].device).synchronize()
        self.hub.barrier.wait()  # (nobody overwrites its send blocks while another rank still copies from them)


def on_device_rows(fab):
    """does this fabric keep the row blocks on a GPU (then the windows' rows compete for its HBM)"""
    return getattr(fab, "device", None) is not None and fab.device.type == "cuda"


def deal_sections(sections, rank, world):
    """the sections a rank owns: round-robin in the given order (every rank computes the same deal)"""
    return [s for i, s in enumerate(sections) if i % world == rank]


def run_chunk_by_targets(out_dir, chunk_index=0, painting=None, device=None, sections=None, in_flight=4,
                         build_on_gpu=True, window_rows=None, sum_mode=0, from_paint_files=False, no_consistency=False,
                         fb=0, fabric=None, shard=None, num_sections=None, N=None, idle_sleep=0.001):  # noqa: C901
    """Paint -> BuildTopology of ONE chunk too large for one GPU, sharded by target haplotype (BASELINE.json config #5):
    this rank paints target_range(rank, world, N), owns deal_sections(sections) -- at most `in_flight` at a time, a host
    thread each -- and serves its rows of every matrix any owner asks for (see the protocol above).  Writes
    <out_dir>/chunk_<c>/<out>_<section>.anc/.mut for the sections it owns, the files rl_stage_build_topology writes.
    Returns {section: number of trees} for those.

    fabric: TorchFabric (default: the torch.distributed job, one process per GPU) or a ThreadFabric.  shard: an object
    with api.Shard's methods (default: api.Shard for this rank's range; tests pass stand-ins).  window_rows: posterior
    rows a window keeps resident (None = from the free HBM, 0 = all; a rank holds in_flight x world windows of its
    share of the targets)."""
    import ctypes as C
    import threading
    import time
    import numpy as np

    fab = fabric if fabric is not None else TorchFabric(device)
    rank, world = fab.rank, fab.world
    dev = device if device is not None else local_device()
    if shard is None:
        from relate_amd import api
        if N is None:
            N = int(np.fromfile(os.path.join(out_dir, "parameters_c%d.bin" % chunk_index), dtype=np.int32, count=1)[0])
        k0, k1 = target_range(rank, world, N)
        shard = api.Shard(out_dir, chunk_index, k0, k1, painting=painting, sum_mode=sum_mode, device=dev,
                          from_paint_files=from_paint_files)
        owns_shard = True
    else:
        owns_shard = False
    N = shard.N
    if (shard.k_begin, shard.k_end) != target_range(rank, world, N):
        raise ValueError("rank %d of %d holds targets %d..%d, not %s" % (rank, world, shard.k_begin, shard.k_end,
                                                                          target_range(rank, world, N)))
    if window_rows is None:
        # from the HBM that is free once the shard has painted: this rank serves in_flight x world windows at a time
        # (its targets' rows of every section in flight anywhere), a posterior row is ~4.2 N bytes, and the trees of
        # the sections it owns need ~17 N^2 bytes each on the device; 0.7 of what is left for the rows
        window_rows = 0
        if on_device_rows(fab):
            nloc = shard.k_end - shard.k_begin
            n_win = max(1, int(in_flight)) * world
            mine_n = min(max(1, int(in_flight)), max(1, len(deal_sections(
                list(range(shard.W if num_sections is None else num_sections)) if sections is None else list(sections),
                rank, world))))
            free_b = float(torch.cuda.mem_get_info(fab.device)[0])
            row_b = 4.2 * N
            # next to its rows a bounded window keeps one forward and one backward RePaint state per target (doubles:
            # 4 rows' worth of bytes each) and its slice of the stepping stones (2 N floats per target); RePaint's
            # strips for a window's first, whole pass are one buffer of the context (a sixth of the rows, doubles)
            per_window = 4.0 * row_b * nloc + 8.0 * N * nloc + 48e6
            avail = 0.85 * free_b - (17.0 * N * N * mine_n if build_on_gpu else 0.0) - n_win * per_window
            min_rows = 3 * nloc + 64
            if avail < n_win * min_rows * row_b:
                raise ValueError("run_chunk_by_targets: %d windows in flight (in_flight %d x %d ranks) do not fit the %.0f GB "
                                 "free on the device next to %d tree builders -- fewer in flight" %
                                 (n_win, in_flight, world, free_b / 1e9, mine_n))
            # (a third of what is left goes to the strips of the whole passes)
            window_rows = max(min_rows, int(0.67 * avail / (n_win * row_b)))
    if window_rows:
        shard.set_window_rows(window_rows)
    todo = list(range(shard.W if num_sections is None else num_sections)) if sections is None else list(sections)
    mine = deal_sections(todo, rank, world)
    Q = max(1, int(in_flight))
    maxrows = -(-N // world)
    # a block = this rank's rows + one row whose first float says whether they are good (1.0 = shard.rows failed on
    # that rank and the block is stale: the owner must not build from it)
    stride = maxrows + 1
    # the matrices of a tick are exchanged `lanes` at a time: every rank computes its rows of up to `lanes` requests,
    # the collectives queue up behind one another and ONE wait covers them (a wait per matrix serialised ~1.5 ms of
    # exchange per tree; two ranks as threads of one process paid two barriers per matrix)
    lanes = max(1, min(4, Q * world))
    sends = [fab.buffer(stride, N) for _ in range(lanes)]
    recvs = [fab.buffer(world * stride, N) for _ in range(lanes)]
    on_device = sends[0].is_cuda
    gpu_build = bool(build_on_gpu and on_device)
    if gpu_build:
        shard.expect_builders(min(Q, max(1, len(mine))))

    lock = threading.Lock()
    pending = [None] * Q          # slot -> [kind, section, snp, ptr, to_device, event, error]
    results, failures = {}, []
    next_mine = [0]
    active = [0]                  # owner threads still running

    stopped = []                  # the job's error once the tick loop has ended on one

    def post(slot, kind, section, snp=-1, ptr=0, to_device=False):
        ev = threading.Event()
        req = [kind, section, snp, ptr, to_device, ev, None]
        with lock:
            if stopped:
                if kind == REQ_RELEASE:
                    return
                raise stopped[0]
            pending[slot] = req
        ev.wait()
        if req[6] is not None:
            raise req[6]

    def owner(slot):
        try:
            while True:
                with lock:
                    if next_mine[0] >= len(mine) or failures:
                        return
                    section = mine[next_mine[0]]
                    next_mine[0] += 1
                try:
                    results[section] = shard.build_section(
                        section, matrix=lambda snp, ptr: post(slot, REQ_MATRIX, section, snp, ptr, False),
                        matrix_dev=(lambda snp, ptr: post(slot, REQ_MATRIX, section, snp, ptr, True)) if gpu_build else None,
                        build_device=dev if gpu_build else None, no_consistency=no_consistency, fb=fb)
                finally:
                    post(slot, REQ_RELEASE, section)
        except BaseException as e:
            with lock:
                failures.append(e)
        finally:
            with lock:
                active[0] -= 1

    threads = [threading.Thread(target=owner, args=(q,), daemon=True) for q in range(min(Q, max(1, len(mine))))]
    active[0] = len(threads)
    for t in threads:
        t.start()

    def deliver(req, recv):
        _, _, _, ptr, to_device, _, _ = req
        for r in range(world):
            a, b = target_range(r, world, N)
            if b == a:
                continue
            blk = recv[r * stride: r * stride + (b - a)]
            if to_device:
                shard.copy_on_device(ptr + a * N * 4, blk.data_ptr(), (b - a) * N * 4)
            else:
                dst = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_float)), shape=(N, N))
                dst[a:b] = blk.cpu().numpy()

    served = 0
    abort = None
    while True:
        table = np.zeros((Q + 1, 3), dtype=np.int64)
        with lock:
            snapshot = list(pending)
            idle = active[0] == 0 and all(p is None for p in snapshot)
            failed = bool(failures) or abort is not None
        for q, req in enumerate(snapshot):
            if req is not None:
                table[q] = (req[0], req[1], req[2])
        table[Q] = (-1 if failed else (1 if idle else 0), 0, 0)
        tables = fab.all_gather_table(table)
        if (tables[:, Q, 0] < 0).any():  # some rank failed: everybody stops in the same tick
            with lock:
                err = abort or (failures[0] if failures else RuntimeError("run_chunk_by_targets: another rank failed"))
                failures.insert(0, err)
                stopped.append(err)
                for q, req in enumerate(pending):
                    if req is not None:
                        req[6] = err
                        pending[q] = None
                        req[5].set()
            for t in threads:
                t.join(timeout=60)
            if owns_shard:
                shard.close()
            raise err
        # the tick's requests in (rank, slot) order; runs of matrix requests go `lanes` at a time
        reqs = []
        for r in range(world):
            for q in range(Q):
                kind, section, snp = (int(x) for x in tables[r, q])
                if kind != REQ_NONE:
                    reqs.append((r, q, kind, section, snp))
        busy = bool(reqs)

        def complete(r, q, error=None):
            if r == rank:
                req = snapshot[q]
                if error is not None:
                    req[6] = error
                with lock:
                    pending[q] = None
                req[5].set()
        i = 0
        while i < len(reqs):
            r, q, kind, section, snp = reqs[i]
            if kind == REQ_RELEASE:
                try:
                    shard.release_section(section)
                except BaseException as e:
                    abort = e
                complete(r, q)
                i += 1
                continue
            group = []
            while i < len(reqs) and reqs[i][2] == REQ_MATRIX and len(group) < lanes:
                group.append(reqs[i])
                i += 1
            for lane, (r, q, kind, section, snp) in enumerate(group):
                sends[lane][maxrows, 0] = 0.0
                try:
                    if abort is None:
                        shard.rows(section, snp, sends[lane].data_ptr())
                    else:
                        sends[lane][maxrows, 0] = 1.0  # (nothing computed: the block is stale)
                except BaseException as e:  # (keep the collectives aligned; the next tick stops the job)
                    abort = e
                    sends[lane][maxrows, 0] = 1.0
            fab.all_gather_rows_many(sends[:len(group)], recvs[:len(group)])
            served += len(group)
            for lane, (r, q, kind, section, snp) in enumerate(group):
                # (the ok-flag rows: only the rank that CONSUMES the matrix looks at them -- a device-to-host copy per tree
                #  it owns; its `abort` reaches the others through the next tick's table)
                if r == rank and abort is None:
                    bad = [int(x) for x in torch.nonzero(recvs[lane][maxrows::stride, 0].cpu()).flatten()]
                    if bad:
                        abort = RuntimeError("run_chunk_by_targets: rows of section %d at SNP %d failed on rank%s %s"
                                             % (section, snp, "s" if len(bad) > 1 else "", bad))
                if r == rank and abort is None:
                    deliver(snapshot[q], recvs[lane])
                    complete(r, q)
                else:
                    complete(r, q, abort)  # (no tree from a matrix with stale rows in it)
        if not busy:
            if (tables[:, Q, 0] == 1).all():
                break
            time.sleep(idle_sleep)
    for t in threads:
        t.join()
    if gpu_build:
        shard.expect_builders(0)
    if owns_shard:
        shard.close()
    if failures:
        raise failures[0]
    return results


def main(argv=None):
    """`python -m torch.distributed.run --nproc-per-node G -m relate_amd.dist OUT_DIR [--painting theta,rho]`:
    every chunk of OUT_DIR/parameters.bin through Paint, BuildTopology and FindEquivalentBranches, chunk c on rank
    c mod G (RelateParallel.sh:216-262 for one process per GPU).  Without a launcher it is one rank on the local GPU.
    `--by-targets`: every chunk on ALL ranks instead, sharded by target haplotype (config #5, run_chunk_by_targets)."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m relate_amd.dist")
    ap.add_argument("out_dir")
    ap.add_argument("--painting", default=None, help="theta,rho (Relate's --painting)")
    ap.add_argument("--chunks", default=None, help="comma-separated subset of chunk indices")
    ap.add_argument("--paint-files", dest="paint_files", action="store_true",
                    help="Paint and BuildTopology as two stages with the paint files in between (the reference's route)")
    ap.add_argument("--by-targets", dest="by_targets", action="store_true",
                    help="every chunk sharded by TARGET haplotype over all ranks (BASELINE.json config #5: a chunk whose "
                         "stepping stones do not fit one GPU; run_chunk_by_targets) instead of chunk c on rank c mod G")
    ap.add_argument("--in-flight", dest="in_flight", type=int, default=4, help="--by-targets: sections a rank owns at once")
    ap.add_argument("--window-rows", dest="window_rows", type=int, default=None,
                    help="--by-targets: posterior rows a window keeps resident per rank (default: from the free HBM; 0: all)")
    ap.add_argument("--stages", default=None,
                    help="importable module with stage_paint, stage_build_topology, stage_paint_build_topology, "
                         "stage_find_equivalent_branches, num_sections (default: relate_amd.api, the GPU library) -- e.g. "
                         "wrappers around another Relate binary")
    args = ap.parse_args(argv)
    painting = tuple(float(x) for x in args.painting.split(",")) if args.painting else None
    chunks = [int(x) for x in args.chunks.split(",")] if args.chunks else None
    launched = "RANK" in os.environ
    if launched:
        import datetime
        import torch
        on_gpu = torch.cuda.is_available()
        if on_gpu:
            torch.cuda.set_device(local_device())
        # (a rank dealt one chunk fewer waits for the others at the end: a chunk's stages can take longer than the
        #  backend's default timeout)
        dist.init_process_group("nccl" if on_gpu else "gloo", timeout=datetime.timedelta(hours=24))
    stages = None
    if args.stages:
        import importlib
        stages = importlib.import_module(args.stages)
    if args.by_targets:
        rank = dist.get_rank() if launched else 0
        world = dist.get_world_size() if launched else 1
        todo = list(range(read_parameters(args.out_dir)["num_chunks"])) if chunks is None else chunks
        for c in todo:
            if args.paint_files:
                # the reference's route: the Paint stage writes the chunk's paint files (rank 0, all targets), the
                # shards read their targets' records from them
                if rank == 0:
                    (stages if stages is not None else __import__("relate_amd.api", fromlist=["api"])).stage_paint(
                        args.out_dir, c, painting=painting, device=local_device())
                if launched:
                    dist.barrier()
            shard_obj = None
            if stages is not None and hasattr(stages, "Shard"):  # (a stand-in for api.Shard: tests, other back ends)
                import inspect
                import numpy as np
                N = int(np.fromfile(os.path.join(args.out_dir, "parameters_c%d.bin" % c), dtype=np.int32, count=1)[0])
                kw = {}
                accepted = inspect.signature(stages.Shard).parameters
                for name, val in (("painting", painting), ("from_paint_files", args.paint_files)):
                    if name in accepted:
                        kw[name] = val
                shard_obj = stages.Shard(args.out_dir, c, *target_range(rank, world, N), **kw)
            try:
                owned = run_chunk_by_targets(args.out_dir, c, painting=painting, in_flight=args.in_flight,
                                             window_rows=args.window_rows, from_paint_files=args.paint_files,
                                             shard=shard_obj)
            finally:
                if shard_obj is not None and hasattr(shard_obj, "close"):
                    shard_obj.close()  # (run_chunk_by_targets closes only the shard it opened itself)
            print("rank %d of %d: chunk %d by targets, owned sections %s" % (rank, world, c, sorted(owned)), flush=True)
            if launched:
                dist.barrier()  # (every section's files are written before the host-only stage reads them)
            if rank == 0:
                (stages if stages is not None else __import__("relate_amd.api", fromlist=["api"])).stage_find_equivalent_branches(
                    args.out_dir, c)
        mine = todo
    else:
        mine = run_chunks(args.out_dir, painting=painting, chunks=chunks, paint_files=args.paint_files, stages=stages)
    print("rank %d ran chunks %s" % (dist.get_rank() if launched else 0, mine), flush=True)
    if launched:
        dist.barrier()  # (the job ends together; nothing is exchanged)
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
