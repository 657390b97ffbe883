#!/usr/bin/env python3
"""bench.py -- decoded codewords/s of the batched min-sum decoder on MI355X.

Metric (BASELINE.json): decoded codewords/sec @ 25 min-sum iterations, TM8192, Eb/N0 = 2 dB.
A "step" is one pass of the hot path (one decode_ms kernel launch) over this rank's shard of
synthetic AWGN frames, already resident in HBM.  Independent frames shard across GPUs with no
collective (weak scaling: every GPU decodes --frames-per-gpu frames; at 8 GPUs the default is
exactly BASELINE config 4, 4 194 304 frames).  torch is used for device memory, streams,
events and the cross-rank barrier only.

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  roofline     -- algorithmic HBM bytes per launch / measured kernel time vs the 8 TB/s peak
  cpu_baseline -- the CPU oracle (a C port of the reference; Rust cannot be built here) timed
                  on this host's cores on a bounded sample of the same frames, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes_per_codeword(code, itemsize):
    # n*sizeof(T) LLRs in + (n+p)/8 hard bits out + 5 status bytes
    return code.n() * itemsize + code.output_len() + 5


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--code", default="TM8192")
    ap.add_argument("--dtype", default="f32", choices=["f32", "i8"])
    ap.add_argument("--ebn0", type=float, default=2.0)
    ap.add_argument("--maxiters", type=int, default=25)
    ap.add_argument("--frames-per-gpu", type=int, default=524288)
    ap.add_argument("--pool", type=int, default=256, help="distinct random codewords the frames cycle through")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from labrador_ldpc_amd import LDPCCode
    from labrador_ldpc_amd.sharding import frame_seed, reduce_max

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decoder has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    code = LDPCCode[args.code]
    F = args.frames_per_gpu
    itemsize = 4 if args.dtype == "f32" else 1
    rate = code.k() / code.n()
    sigma = float(np.sqrt(1.0 / (2.0 * rate * 10.0 ** (args.ebn0 / 10.0))))

    # ---- synthetic frames: random codewords (product's host encoder) -> BPSK+AWGN on the device
    rng = np.random.default_rng(0x1DBC + int(code))
    pool = np.zeros((args.pool, code.n() // 8), dtype=np.uint8)
    for i in range(args.pool):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    d_pool = torch.from_numpy(pool).to(dev)
    seed = frame_seed(0x1DBC + int(code), rank)
    llrs = code.awgn_frames(d_pool, F, sigma, seed, dtype=args.dtype)
    out = torch.empty((F, code.output_len()), dtype=torch.uint8, device=dev)
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    succ = torch.empty((F,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        code.decode_ms_batch(llrs, args.maxiters, output=out, iters=iters, success=succ, variant=args.variant)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    elapsed, kernel_ms_max = reduce_max([elapsed, kernel_ms], device=dev)   # slowest rank

    mean_iters = float(iters.double().mean())
    frame_fail = 1.0 - float(succ.double().mean())

    result = None
    if rank == 0:
        value = world * F * args.steps / elapsed
        bytes_per_launch = F * algorithmic_bytes_per_codeword(code, itemsize)
        achieved = bytes_per_launch / (kernel_ms_max * 1e-3) / 1e9
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    t = json.load(f)
                key = f"{args.code}_{args.dtype}"
                if key in t and t[key].get("frames"):
                    traffic = t[key]["hbm_bytes_per_launch"] * (F / t[key]["frames"])
                    if t[key].get("valu_insts_per_launch"):
                        # The limiter this kernel actually runs into: wave-level VALU instructions
                        # (PMC SQ_INSTS_VALU of the profiled launch, scaled to this one) against the issue peak of
                        # 256 CUs x 4 SIMDs x 2 wave-instructions per 4 cycles at 2.4 GHz.
                        insts = t[key]["valu_insts_per_launch"] * (F / t[key]["frames"])
                        peak = 256 * 4 * 2 / 4 * 2.4                                  # G wave-instructions/s
                        ach = insts / (kernel_ms_max * 1e-3) / 1e9
                        valu = {"achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                                "source": "SQ_INSTS_VALU per frame from profiles/ (same code, Eb/N0 and iteration cap)"}
            except Exception:
                traffic = None
        result = {
            "metric": "decoded codewords/sec @25 min-sum iters, TM8192, Eb/N0=2dB; 1/2/4/8 GPU" if
                      (args.code == "TM8192" and args.maxiters == 25 and args.ebn0 == 2.0) else
                      f"decoded codewords/sec @{args.maxiters} min-sum iters, {args.code}, Eb/N0={args.ebn0}dB",
            "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.code} {args.dtype} LLRs, {F} frames per GPU resident in HBM "
                                   f"({world * F} total; BASELINE config 4 = 4194304 frames over 8 GPUs), "
                                   f"max_iters {args.maxiters} with early termination, AWGN Eb/N0 {args.ebn0} dB, "
                                   f"{args.pool} random codewords",
                       "code": args.code, "frames_per_gpu": F, "max_iters": args.maxiters, "ebn0_db": args.ebn0,
                       "sigma": sigma, "parallelism": f"{world} independent shard(s), no collective",
                       "kernel_variant": args.variant},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("decode_ms_pair_kernel" if (args.code == "TM8192" and args.dtype != "f64" and args.variant in (0, 32)) else "decode_ms_kernel"), "kernel_ms": kernel_ms_max,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "valu_issue": valu,
            "diag": {"mean_iters_returned": mean_iters, "frame_failure_rate": frame_fail,
                     "edge_visits_per_s": world * F * args.steps / elapsed * 2 * code.paritycheck_sum() * (mean_iters + 1)},
        }

    # ---- CPU baseline + parity on a bounded sample (rank 0, single GPU only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle                                  # test infrastructure: the checker / CPU baseline
        cores = usable_cores()
        probe = min(F, 2 * cores)
        h_llrs = llrs[:probe].cpu().numpy()
        t1 = time.perf_counter()
        oracle.decode_ms_batch(code, h_llrs, args.maxiters, cores)
        per_frame = (time.perf_counter() - t1) / probe
        sample = int(max(probe, min(F, args.cpu_seconds / max(per_frame, 1e-9))))
        sample = min(sample, 65536)
        h_llrs = llrs[:sample].cpu().numpy()
        t1 = time.perf_counter()
        o_c, it_c, ok_c, used = oracle.decode_ms_batch(code, h_llrs, args.maxiters, cores)
        cpu_s = time.perf_counter() - t1
        o_g, it_g, ok_g = out[:sample].cpu().numpy(), iters[:sample].cpu().numpy(), succ[:sample].cpu().numpy()
        mism = int(((it_g.astype(np.int64) != it_c.astype(np.int64)) | (ok_g != ok_c) | (o_g != o_c).any(axis=1)).sum())
        result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "codewords/s", "cores": used, "kind": "port",
                                  "sample": f"first {sample} of the {F} frames the GPU decoded, same LLR bits, "
                                            f"{cpu_s:.1f} s wall on {used} threads (C port of decode_ms, gcc -O3; "
                                            f"the Rust reference cannot be built in this image)"}
        result["parity"] = {"frames_compared": sample, "mismatches": mism}
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
