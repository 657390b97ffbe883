"""Multi-GPU sharding of a batch of independent frames.

Codewords are independent (the reference keeps no state across calls, src/lib.rs:15-17), so a
batch splits into contiguous per-rank slices with NO data-path collective; the only cross-rank
traffic is the timing/counter reduction below (what perftest aggregates through one AtomicU64,
perftest/src/main.rs:43).  Backend-agnostic: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

from typing import Sequence, Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(start, count) of rank's contiguous slice of `total` frames; slices differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def frame_seed(base_seed: int, rank: int) -> int:
    """Per-rank 64-bit seed: frames of different ranks use disjoint generator streams."""
    return (base_seed & 0xFFFFFFFFFF) | ((rank & 0xFFFFFF) << 40)


def reduce_max(values: Sequence[float], device=None) -> Sequence[float]:
    """Element-wise MAX over ranks (elapsed time of the slowest rank); identity without a group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def reduce_sum(values: Sequence[float], device=None) -> Sequence[float]:
    """Element-wise SUM over ranks (frames, failures, iterations)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
