"""Multi-GPU plumbing: streams shard by rank, nothing else is exchanged.

Streams are independent, so the data path has NO collective: every rank owns a
contiguous block of streams with a replicated model.  The only communication is the
end-of-run reduction of (elapsed, frames) -- `torch.distributed` over RCCL on GPUs (backend
"nccl"), gloo in the CPU tests.
"""
from __future__ import annotations


def shard_streams(total_streams: int, world_size: int, rank: int) -> range:
    """Contiguous, balanced partition of stream ids 0..total-1 (sizes differ by at most 1)."""
    base, extra = divmod(total_streams, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def aggregate_throughput(frames: float, elapsed: float, dist=None, device=None):
    """(total frames over ranks, max elapsed over ranks).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(frames), float(elapsed)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    f = torch.tensor([frames], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(f.item()), float(t.item())


def aggregate_times(times, dist=None, device=None, group=None):
    """Element-wise MAX over ranks of a list of per-repetition elapsed times (one all-reduce on `group`, default: the job's)."""
    times = [float(t) for t in times]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return times
    import torch
    t = torch.tensor(times, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.tolist()]
