"""Multi-GPU sharding of a batch of independent frames.

Codewords are independent (the reference keeps no state across calls, src/lib.rs:15-17), so a
batch splits into contiguous per-rank slices with NO data-path collective; the only cross-rank
traffic is the timing/counter reduction below (what perftest aggregates through one AtomicU64,
perftest/src/main.rs:43), done host-side over a gloo group: the path needs no RCCL.

One process per GPU.  `spawn_local_ranks` starts them when no launcher did (bench.py --gpus N);
under `python -m torch.distributed.run` the launcher's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are
used as they are.  This module imports nothing that touches a GPU."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Sequence, Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(start, count) of rank's contiguous slice of `total` frames; slices differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def frame_seed(base_seed: int, rank: int) -> int:
    """Per-rank 64-bit seed: frames of different ranks use disjoint generator streams.  (bench.py no longer uses it: since
    round 4 frames are keyed by their GLOBAL index in the job -- labrador_ldpc_hip_awgn_*_at -- so that a shard is a slice of
    the one-GPU batch bit for bit; the per-rank seed remains for harnesses that want independent streams per worker.)"""
    return (base_seed & 0xFFFFFFFFFF) | ((rank & 0xFFFFFF) << 40)


def spawn_local_ranks(argv: List[str], n: int, env: Optional[Dict[str, str]] = None, timeout: Optional[float] = None) -> int:
    """Run `python argv...` as n rank processes of one node (RANK = LOCAL_RANK = 0..n-1, WORLD_SIZE = n,
    rendezvous on 127.0.0.1 at a free port) and return the worst exit status.  The caller must not have
    initialised a GPU; the children are fresh interpreters.  If a rank fails the others are stopped
    (by PID)."""
    if n < 1:
        raise ValueError("n must be >= 1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    import time
    deadline = None if timeout is None else time.monotonic() + timeout
    worst, live = 0, list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 1)
                for q in live:                   # one rank failed: the others would wait at the barrier for ever
                    q.terminate()
        if deadline is not None and time.monotonic() > deadline:
            for q in live:
                q.kill()
            worst = worst or 124
            deadline = None
        time.sleep(0.05)
    return worst


def init_ranks() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the launcher's environment; with world > 1 joins the gloo group
    used for the barrier and the timing reduction (CPU tensors: no device, no RCCL)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            # gloo announces its connections on STDOUT from C++; the caller's stdout carries exactly one JSON
            # line (bench.py), so the file descriptor is pointed at stderr for the duration of the rendezvous
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()                    # the peers' connection messages appear here
            finally:
                os.dup2(saved, 1)
                os.close(saved)
    return rank, local_rank, world


def barrier() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def finish_ranks() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


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


def gather(values: Sequence[float]) -> List[List[float]]:
    """Every rank's `values` (equal lengths), indexed by rank, on every rank: per-rank diagnostics of a bench line (device,
    kernel time, clock) -- what tells a slow GPU from a slow shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.float64)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def reduce_sum_int(values: Sequence[int]) -> Sequence[int]:
    """Element-wise EXACT integer SUM over ranks (int64; the float64 of reduce_sum is exact only to 2^53): iteration totals,
    failure counts and the job digest of bench.py, which must be equal whatever the number of ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
