"""The N>1 path on CPU: two gloo ranks each decode their own contiguous shard (with the oracle
standing in for the GPU kernel), no data-path collective; only timing/counters are reduced."""
import os
import socket

import numpy as np
import pytest

from labrador_ldpc_amd.sharding import frame_seed, reduce_max, reduce_sum, shard_range


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 4194304):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert shard_range(4194304, 8, 3) == (3 * 524288, 524288)      # BASELINE config 4
    assert len({frame_seed(0x1DBC + 8, r) for r in range(8)}) == 8


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        code = 2                                               # TC512
        start, count = shard_range(total, world, rank)
        rng = np.random.default_rng(frame_seed(1234, rank))
        llrs, _ = oracle.awgn_llrs(code, rng, count, 3.0, np.float32)
        out, iters, ok, _ = oracle.decode_ms_batch(code, llrs, 25, 1)
        elapsed = 1.0 + rank                                   # pretend rank 1 is slower
        mx = reduce_max([elapsed])
        sm = reduce_sum([count, float((ok == 0).sum()), float(iters.sum())])
        q.put((rank, start, count, mx[0], sm, float(llrs[0, 0])))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shards():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 37
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, c0, mx0, sm0, x0), (r1, s1, c1, mx1, sm1, x1) = res
    assert (s0, c0, s1, c1) == (0, 19, 19, 18)                 # contiguous, disjoint, complete
    assert mx0 == mx1 == 2.0                                   # slowest rank's time on every rank
    assert sm0 == sm1 and sm0[0] == total                      # counters summed over ranks
    assert x0 != x1                                            # different generator streams
