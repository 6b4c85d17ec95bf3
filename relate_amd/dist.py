"""Multi-GPU partitioning of the Paint -> BuildTopology path.

The path shards by genomic chunk (one process per GPU, no data-path
collective): chunks are independent after MakeChunks, exactly as the
reference's cluster scripts run them (scripts/RelateParallel/RelateParallel.sh:216,
scripts/RelateSGE/RelateSGE.sh:324-401).  Within a chunk, BuildTopology shards
by section (window) the same way (RelateParallel.sh:231-257).  The only
collectives are the bookkeeping ones below (job statistics)."""
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
