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


def _digest_worker(rank, world, port, total, q):
    """One rank of a job whose frames are keyed by their GLOBAL index (what bench.py does on the GPU since round 4): the shard is a
    slice of the one-process job, and the exact integer results sum over the ranks to the one-process job's."""
    import importlib.util
    import torch
    import torch.distributed as dist
    import oracle
    from labrador_ldpc_amd.sharding import reduce_sum_int
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(root, "bench.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        code = 2                                               # TC512
        llrs, _ = oracle.awgn_llrs(code, np.random.default_rng(77), total, 2.0, np.float32)     # THE job: every rank can name any frame of it
        start, count = shard_range(total, world, rank)
        out, iters, ok, _ = oracle.decode_ms_batch(code, llrs[start:start + count], 25, 1)

        class W(b.Workload):
            def __init__(self):
                self.first_frame, self.frames = start, count
                self.out, self.iters, self.succ = torch.from_numpy(out), torch.from_numpy(iters.astype(np.int32)), torch.from_numpy(ok.astype(np.uint8))
        it_sum, fails, digest = reduce_sum_int(list(W().sums()))
        q.put((rank, it_sum, fails, digest % b.DIGEST_MOD))
    finally:
        dist.destroy_process_group()


def test_job_results_are_equal_for_one_and_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    seen = []
    for world in (1, 2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_digest_worker, args=(r, world, port, 41, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert len({r[1:] for r in res}) == 1                   # every rank holds the same job-wide sums
        seen.append(res[0][1:])
    assert seen[0] == seen[1] and seen[0][0] > 0, seen           # ... and the 2-rank job IS the 1-rank job


PROBE = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
from labrador_ldpc_amd.sharding import init_ranks, reduce_max, reduce_sum, gather, barrier, finish_ranks, shard_range
rank, local_rank, world = init_ranks()
if len(sys.argv) > 3 and int(sys.argv[3]) == rank:
    sys.exit(7)                                   # a failing rank, before the group's first collective
barrier()
start, count = shard_range(int(sys.argv[2]), world, rank)
mx = reduce_max([1.0 + rank, 10.0 - rank])
sm = reduce_sum([count])
per_rank = gather([float(rank), float(start), float(count)])        # bench.py's per-rank diagnostics: every rank's row, by rank
if rank == 0:
    print(json.dumps({"world": world, "local_rank": local_rank, "max": mx, "frames": sm[0], "ranks": per_rank}), flush=True)
finish_ranks()
'''


def test_spawn_local_ranks_runs_one_process_per_rank(tmp_path, capfd):
    """What `python bench.py --gpus N` does when no launcher set WORLD_SIZE: N fresh rank processes,
    gloo rendezvous on 127.0.0.1, host-side MAX / SUM of the timings and counters, one line from rank 0."""
    import json
    from labrador_ldpc_amd.sharding import spawn_local_ranks
    script = tmp_path / "probe.py"
    script.write_text(PROBE)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    assert spawn_local_ranks([str(script), root, "37"], 2, env=env, timeout=120) == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got == {"world": 2, "local_rank": 0, "max": [2.0, 10.0], "frames": 37.0, "ranks": [[0.0, 0.0, 19.0], [1.0, 19.0, 18.0]]}
    from labrador_ldpc_amd.sharding import gather
    assert gather([3.0, 4.0]) == [[3.0, 4.0]]                       # without a group: this process's row alone
    # a rank that dies takes the job down with a non-zero status instead of hanging the others
    assert spawn_local_ranks([str(script), root, "37", "1"], 2, env=env, timeout=120) != 0


def test_bench_py_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks by itself;
    on a box without GPUs each rank then stops with the no-GPU message -- never with a launcher error."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                       # also on a GPU box this test stays a launcher test
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr
    assert "launch with" not in r.stderr
