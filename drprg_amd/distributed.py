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
    # ONE collective: coverage, per-PRG cluster counts and the base count travel in one int32 buffer --
    # the message is a few hundred KB at most, so the exchange is latency-bound and every extra collective costs a round trip
    tb = int(total_bases)
    n_c, n_p = covg.numel(), prg_reads.numel()
    buf = torch.empty(n_c + n_p + 4, dtype=torch.int32, device=covg.device)
    buf[:n_c] = covg.reshape(-1)
    buf[n_c:n_c + n_p] = prg_reads.reshape(-1)
    # the 64-bit base count as four 16-bit pieces (int32 lanes do not carry into each other)
    pieces = torch.tensor([tb & 0xFFFF, (tb >> 16) & 0xFFFF, (tb >> 32) & 0xFFFF, tb >> 48], dtype=torch.int32, device=covg.device)
    buf[n_c + n_p:] = pieces
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    covg.reshape(-1).copy_(buf[:n_c])
    prg_reads.reshape(-1).copy_(buf[n_c:n_c + n_p])
    p = [int(x) for x in buf[n_c + n_p:].tolist()]  # each piece summed over <= 2^15 ranks stays below 2^31
    return p[0] + (p[1] << 16) + (p[2] << 32) + (p[3] << 48)
