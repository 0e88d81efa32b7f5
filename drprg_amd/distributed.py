"""Multi-GPU plumbing of the hot path: reads shard embarrassingly across ranks (one process per GPU);
the only exchange step is one sum-reduce of the per-k-mer-node coverage vector (RCCL over xGMI when the
process group is `nccl`, gloo on CPU in the tests).  Unsigned integer sums commute, so the reduced
vector -- and everything genotyped from it -- is independent of the number of ranks.
"""
import numpy as np


def shard_range(n_reads, rank, world):
    """Contiguous shard [lo, hi) of reads for `rank` (N/ndev contiguous ranges)."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(bases, offsets, rank, world):
    """Slice a (bases, offsets) batch to this rank's shard, re-basing the offsets to 0."""
    n = len(offsets) - 1
    lo, hi = shard_range(n, rank, world)
    b0, b1 = int(offsets[lo]), int(offsets[hi])
    return bases[b0:b1], (np.asarray(offsets[lo:hi + 1], dtype=np.uint64) - np.uint64(b0))


def reduce_coverage(covg, prg_reads, total_bases, group=None):
    """All-reduce (sum) of the coverage tensors in place; returns the global base count.

    covg / prg_reads are torch int32 tensors (the u32 counters reinterpreted: sums stay below 2^31 for any
    realistic depth; pandora itself saturates at 65535 per k-mer).  With an uninitialised process group
    (single rank) this is a no-op.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(total_bases)
    dist.all_reduce(covg, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(prg_reads, op=dist.ReduceOp.SUM, group=group)
    t = torch.tensor([int(total_bases)], dtype=torch.int64, device=covg.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())
