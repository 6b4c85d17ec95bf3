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


def run_chunks(out_dir, painting=None, device=None, stages=None, chunks=None, paint_files=False):
    """The many-chunks route (BASELINE.json config #4; scripts/RelateParallel/RelateParallel.sh:216-262): the chunks of
    a MakeChunks directory are dealt round-robin to the ranks of the job (one process per GPU) and every rank runs
    its chunks start to end -- Paint + BuildTopology of all sections in ONE stage call with the stepping stones kept
    in HBM (rl_stage_paint_build_topology; paint_files=True takes the reference's two stages with the paint files in
    between: 12.8 GB per C4-sized chunk, eight ranks on one filesystem), then FindEquivalentBranches -- with NO
    data-path collective and no barrier: chunks share nothing but the input directory, a rank is done when its
    chunks are.  Returns the chunk indices this rank ran.  `chunks`: a subset to run (default: all of parameters.bin)."""
    if stages is None:
        from relate_amd import api as stages
    live = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if live else 0
    world = dist.get_world_size() if live else 1
    dev = device if device is not None else local_device()
    todo = list(range(read_parameters(out_dir)["num_chunks"])) if chunks is None else list(chunks)
    mine = shard(todo, rank, world)
    for c in mine:
        last = stages.num_sections(out_dir, c) - 1
        if paint_files:
            stages.stage_paint(out_dir, c, painting=painting, device=dev)
            stages.stage_build_topology(out_dir, c, 0, last, painting=painting, device=dev)
        else:
            stages.stage_paint_build_topology(out_dir, c, 0, last, painting=painting, device=dev)
        stages.stage_find_equivalent_branches(out_dir, c)
    return mine


def main(argv=None):
    """`python -m torch.distributed.run --nproc-per-node G -m relate_amd.dist OUT_DIR [--painting theta,rho]`:
    every chunk of OUT_DIR/parameters.bin through Paint, BuildTopology and FindEquivalentBranches, chunk c on rank
    c mod G (RelateParallel.sh:216-262 for one process per GPU).  Without a launcher it is one rank on the local GPU."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m relate_amd.dist")
    ap.add_argument("out_dir")
    ap.add_argument("--painting", default=None, help="theta,rho (Relate's --painting)")
    ap.add_argument("--chunks", default=None, help="comma-separated subset of chunk indices")
    ap.add_argument("--paint-files", dest="paint_files", action="store_true",
                    help="Paint and BuildTopology as two stages with the paint files in between (the reference's route)")
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
    mine = run_chunks(args.out_dir, painting=painting, chunks=chunks, paint_files=args.paint_files)
    print("rank %d ran chunks %s" % (dist.get_rank() if launched else 0, mine), flush=True)
    if launched:
        dist.barrier()  # (the job ends together; nothing is exchanged)
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
