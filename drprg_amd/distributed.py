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
    if not (covg.is_contiguous() and prg_reads.is_contiguous()) or covg.dtype != torch.int32 or prg_reads.dtype != torch.int32:
        raise ValueError("reduce_coverage works in place: covg / prg_reads must be contiguous int32 tensors")
    tb = int(total_bases)
    n_c, n_p = covg.numel(), prg_reads.numel()
    buf = torch.empty(n_c + n_p + 4, dtype=torch.int32, device=covg.device)
    buf[:n_c] = covg.view(-1)
    buf[n_c:n_c + n_p] = prg_reads.view(-1)
    # the 64-bit base count as four 16-bit pieces (int32 lanes do not carry into each other)
    pieces = torch.tensor([tb & 0xFFFF, (tb >> 16) & 0xFFFF, (tb >> 32) & 0xFFFF, tb >> 48], dtype=torch.int32, device=covg.device)
    buf[n_c + n_p:] = pieces
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    covg.view(-1).copy_(buf[:n_c])
    prg_reads.view(-1).copy_(buf[n_c:n_c + n_p])
    p = [int(x) for x in buf[n_c + n_p:].tolist()]  # each piece summed over <= 2^15 ranks stays below 2^31
    return p[0] + (p[1] << 16) + (p[2] << 32) + (p[3] << 48)


class NativeComm:
    """RCCL communicator made through the C ABI (drprg_hip_comm_*), one rank per process / GPU: what a host without PyTorch
    would use.  `exchange(id_bytes_or_None) -> id_bytes` is whatever channel the host has for handing rank 0's 128-byte id to
    the other ranks (bench.py: a torch.distributed broadcast; a Rust host: its own launcher)."""

    def __init__(self, rank, world, device, exchange):
        import ctypes as C
        from ._lib import lib
        self._lib = lib
        ident = (C.c_uint8 * 128)()
        err = None
        if rank == 0:
            rc = lib.drprg_hip_comm_unique_id(ident)
            if rc != 0:
                err = f"drprg_hip_comm_unique_id: {lib.drprg_hip_last_error(None).decode()} ({rc})"
        # (rank 0 takes part in the exchange even when it has no id to give: the other ranks are waiting in it)
        raw = exchange(bytes(ident) if rank == 0 and err is None else None)
        if raw is None:
            raise RuntimeError(err or "rank 0 could not make a communicator id")
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        comm = C.c_void_p()
        rc = lib.drprg_hip_comm_init_rank(C.byref(comm), world, ident, rank, device)
        if rc != 0:
            raise RuntimeError(f"drprg_hip_comm_init_rank: {lib.drprg_hip_last_error(None).decode()} ({rc})")
        self.handle = comm

    def allreduce(self, ctx, d_covg=None, d_prg_reads=None, stream=None):
        """in-place sum of the context's (or the given) device vectors over the ranks; asynchronous on the stream"""
        rc = self._lib.drprg_hip_allreduce(ctx._h, self.handle, d_covg, d_prg_reads, stream)
        if rc != 0:
            raise RuntimeError(f"drprg_hip_allreduce: {self._lib.drprg_hip_last_error(ctx._h).decode()} ({rc})")

    def close(self):
        if self.handle:
            self._lib.drprg_hip_comm_destroy(self.handle)
            self.handle = None
