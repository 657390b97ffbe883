#!/usr/bin/env python3
"""bench.py -- decoded codewords/s of the batched min-sum decoder on MI355X.

Metric (BASELINE.json): decoded codewords/sec @ 25 min-sum iterations, TM8192, Eb/N0 = 2 dB.
A "step" is one pass of the hot path (one decode_ms kernel launch through the C ABI) over this rank's
shard of synthetic AWGN frames, already resident in HBM.  Independent frames shard across GPUs with no
collective (weak scaling: every GPU decodes --frames-per-gpu frames; at 8 GPUs the default is exactly
BASELINE config 4, 4 194 304 frames).

Launching:  `python bench.py --gpus N` starts the N ranks itself (one child process per GPU, created
before anything in this process touches a GPU); under `python -m torch.distributed.run ... bench.py
--gpus N` (WORLD_SIZE set) the process is one rank.  The ranks synchronise through a gloo group
(barrier + a 2-element MAX of the timings): north_star's path has no data-path collective, so no RCCL.
torch is plumbing only: device memory, streams, events, the barrier.

Prints ONE JSON line on rank 0 with, beyond the contract's fields,
  roofline     -- algorithmic HBM bytes per launch / kernel time (HIP events on the launch stream) vs 8 TB/s
  valu_issue   -- the limiter the kernel actually runs into
  configs      -- N=1 only: BASELINE configs 2, 3 and 5 (both operating points), a few launches each
  cpu_baseline -- N=1 only: the CPU oracle (C port of the reference; Rust cannot be built in this image),
                  built -march=native on this host, on fixed samples of the frames the GPU decoded:
                  1 core and all cores; the same frames are compared bit for bit (parity)
and exits non-zero, with "value": null, if any compared frame differs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_G = 256 * 4 * 2 / 4 * 2.4   # 256 CUs x 4 SIMDs x 2 wave-instructions per 4 cycles x 2.4 GHz = 1228.8 G/s

# BASELINE.json configs beyond the headline one: (key, code, dtype, frames per GPU, Eb/N0 dB)
EXTRA_CONFIGS = [
    ("config2_TC512_f32", "TC512", "f32", 65536, 2.0),
    ("config3_TM2048_f32", "TM2048", "f32", 1048576, 2.0),
    ("config5_TM5120_i8_4dB", "TM5120", "i8", 524288, 4.0),      # per-GPU slice of 4 194 304 frames over 8 GPUs
    ("config5_TM5120_i8_2dB", "TM5120", "i8", 524288, 2.0),      # nothing converges: fixed 25-iteration work
]


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


def sigma_of(code, ebn0_db):
    import numpy as np
    return float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0_db / 10.0))))


class Workload:
    """Synthetic frames of one configuration, resident on `dev`, and their result buffers."""

    def __init__(self, code, dtype, frames, ebn0, pool, seed_rank, dev):
        import numpy as np
        import torch
        from labrador_ldpc_amd.sharding import frame_seed
        self.code, self.dtype, self.frames, self.ebn0 = code, dtype, frames, ebn0
        self.itemsize = 4 if dtype == "f32" else 1
        self.sigma = sigma_of(code, ebn0)
        rng = np.random.default_rng(0x1DBC + int(code))
        cws = np.zeros((pool, code.n() // 8), dtype=np.uint8)
        for i in range(pool):                                          # random codewords: the product's host encoder
            code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), cws[i])
        d_pool = torch.from_numpy(cws).to(dev)
        self.llrs = code.awgn_frames(d_pool, frames, self.sigma, frame_seed(0x1DBC + int(code), seed_rank), dtype=dtype)
        self.out = torch.empty((frames, code.output_len()), dtype=torch.uint8, device=dev)
        self.iters = torch.empty((frames,), dtype=torch.int32, device=dev)
        self.succ = torch.empty((frames,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

    def step(self, maxiters, variant=0):
        self.code.decode_ms_batch(self.llrs, maxiters, output=self.out, iters=self.iters, success=self.succ, variant=variant)

    def timed(self, maxiters, steps, warmup, variant=0, barrier=lambda: None):
        """(wall seconds of `steps` launches, mean kernel ms from HIP events on the launch stream)."""
        import numpy as np
        import torch
        for _ in range(warmup):
            self.step(maxiters, variant)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:                     # decode_ms_batch launches on torch's current stream, where the events sit
            a.record()
            self.step(maxiters, variant)
            b.record()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def stats(self):
        return float(self.iters.double().mean()), 1.0 - float(self.succ.double().mean())

    def compare(self, maxiters, sample, nthreads, lib=None):
        """Decode the first `sample` frames on the CPU oracle; (seconds, threads, mismatching frames)."""
        import numpy as np
        import oracle
        h = self.llrs[:sample].cpu().numpy()
        t1 = time.perf_counter()
        o_c, it_c, ok_c, used = oracle.decode_ms_batch(self.code, h, maxiters, nthreads, lib=lib)
        cpu_s = time.perf_counter() - t1
        o_g = self.out[:sample].cpu().numpy()
        it_g = self.iters[:sample].cpu().numpy().astype(np.int64)
        ok_g = self.succ[:sample].cpu().numpy().astype(np.int64)
        mism = int(((it_g != it_c.astype(np.int64)) | (ok_g != ok_c.astype(np.int64)) | (o_g != o_c).any(axis=1)).sum())
        return cpu_s, used, mism


def kernel_name(code_name, dtype, variant):
    return "decode_ms_pair_kernel" if (code_name == "TM8192" and variant in (0, 32)) else "decode_ms_kernel"


def profile_counters(key):
    """Per-launch PMC figures of the committed profile of this kernel (profiles/hbm_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f).get(key)
        return t if t and t.get("frames") else None
    except Exception:
        return None


def roofline_of(code, code_name, dtype, variant, frames, kernel_ms):
    itemsize = 4 if dtype == "f32" else 1
    bytes_per_launch = frames * algorithmic_bytes_per_codeword(code, itemsize)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    prof = profile_counters(f"{code_name}_{dtype}")
    traffic = prof["hbm_bytes_per_launch"] * (frames / prof["frames"]) if prof else None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": kernel_name(code_name, dtype, variant), "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "limiter": "VALU issue + LDS, not HBM (34 KB of I/O per ~1.1 M edge visits): see valu_issue"}
    valu = None
    if prof and prof.get("valu_insts_per_launch"):
        # wave-level VALU instructions (PMC SQ_INSTS_VALU of the profiled launch, scaled to this one) against the issue peak
        ach = prof["valu_insts_per_launch"] * (frames / prof["frames"]) / (kernel_ms * 1e-3) / 1e9
        valu = {"achieved": ach, "peak": VALU_PEAK_G, "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_G,
                "source": "SQ_INSTS_VALU per frame from profiles/hbm_traffic.json (same code, Eb/N0 and iteration cap)"}
        if prof.get("avg_issue_cycles_per_instruction"):
            # the peak above is for 2-cycle instructions only; this kernel's mix (tools/valu_mix.py: static count over the
            # iteration loop, per-class issue cost from tools/ubench) averages more, so in VALU CYCLES the fraction is
            c = prof["avg_issue_cycles_per_instruction"]
            valu["cycle_weighted"] = {"avg_issue_cycles_per_instruction": c, "frac_of_valu_cycles_at_2.4GHz": ach / VALU_PEAK_G * c / 2.0,
                                      "note": "before the co-issue of F-class instructions with 4-cycle ones of other waves; "
                                              "the chip holds 2.30 GHz under this load (s_memtime against s_memrealtime, tools/kbench.hip stamps), not 2.4"}
    return roof, valu


def run_rank(args):
    import numpy as np
    import torch
    from labrador_ldpc_amd import LDPCCode
    from labrador_ldpc_amd.sharding import init_ranks, reduce_max, barrier, finish_ranks

    rank, local_rank, world = init_ranks()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decoder has no CPU path)")
    # one GPU per rank: local rank r uses device r, unless --devices maps ranks to ordinals explicitly (testing
    # the N > 1 path on a box with fewer GPUs: "--gpus 2 --devices 0,0")
    ordinal = int(args.devices.split(",")[local_rank]) if args.devices else local_rank
    masked = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if not args.devices and masked and torch.cuda.device_count() == 1:
        ordinal = 0                       # a launcher that gives every rank its own visible device
    if ordinal >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: device {ordinal} requested, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(ordinal)
    dev = torch.device("cuda", ordinal)

    code = LDPCCode[args.code]
    F = args.frames_per_gpu
    w = Workload(code, args.dtype, F, args.ebn0, args.pool, rank, dev)
    elapsed, kernel_ms = w.timed(args.maxiters, args.steps, args.warmup, args.variant, barrier)
    elapsed, kernel_ms_max = reduce_max([elapsed, kernel_ms])              # slowest rank
    mean_iters, frame_fail = w.stats()

    result, failed = None, False
    if rank == 0:
        headline = args.code == "TM8192" and args.maxiters == 25 and args.ebn0 == 2.0 and args.dtype == "f32"
        roof, valu = roofline_of(code, args.code, args.dtype, args.variant, F, kernel_ms_max)
        value = world * F * args.steps / elapsed
        result = {
            "metric": "decoded codewords/sec @25 min-sum iters, TM8192, Eb/N0=2dB; 1/2/4/8 GPU" if headline else
                      f"decoded codewords/sec @{args.maxiters} min-sum iters, {args.code} {args.dtype}, Eb/N0={args.ebn0}dB",
            "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.code} {args.dtype} LLRs, {F} frames per GPU resident in HBM ({world * F} total"
                                   + ("; BASELINE config 4 = 4194304 frames over 8 GPUs" if headline else "")
                                   + f"), max_iters {args.maxiters} with early termination, AWGN Eb/N0 {args.ebn0} dB, "
                                     f"{args.pool} random codewords",
                       "code": args.code, "frames_per_gpu": F, "max_iters": args.maxiters, "ebn0_db": args.ebn0,
                       "sigma": w.sigma, "parallelism": f"{world} independent shard(s), one process per GPU, no collective",
                       "kernel_variant": args.variant},
            "roofline": roof, "valu_issue": valu,
            "diag": {"mean_iters_returned": mean_iters, "frame_failure_rate": frame_fail,
                     "edge_visits_per_s": value * 2 * code.paritycheck_sum() * (mean_iters + 1)},
        }

    # ---- N=1 only: CPU baseline + parity on fixed samples, then the other BASELINE configs --------------
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle                                  # test infrastructure: the checker / CPU baseline
        try:
            native, build = oracle._load(oracle.build_native()), "gcc -O3 -march=native, built on this host"
        except Exception as e:                         # no compiler on the box: the portable build, and say so
            native, build = None, f"gcc -O3 -march=x86-64-v2 (native build failed: {e})"
        cores = usable_cores()
        # fixed sample sizes (runs are comparable): 16384 / 1024 TM8192-sized frames for all cores / one core
        scale = 8192 / code.n()
        s_all, s_one = min(F, int(args.cpu_frames * scale)), min(F, int(args.cpu_frames * scale) // 16)
        t_all, used, mism = w.compare(args.maxiters, s_all, cores, native)
        t_one, _, mism1 = w.compare(args.maxiters, s_one, 1, native)
        result["cpu_baseline"] = {
            "value": s_all / t_all, "unit": "codewords/s", "cores": used, "kind": "port",
            "one_core": {"value": s_one / t_one, "unit": "codewords/s", "cores": 1, "sample": f"first {s_one} frames, {t_one:.1f} s"},
            "sample": f"first {s_all} of the {F} frames the GPU decoded, same LLR bits, {t_all:.1f} s wall on {used} threads "
                      f"(C port of decode_ms, {build}; the Rust reference cannot be built in this image)"}
        result["parity"] = {"frames_compared": s_all, "mismatches": mism + mism1}
        failed = failed or (mism + mism1) != 0
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0 and world == 1 and not args.no_configs:
        del w
        torch.cuda.empty_cache()
        configs = {}
        for key, cname, dtype, frames, ebn0 in EXTRA_CONFIGS:
            c = LDPCCode[cname]
            wl = Workload(c, dtype, frames, ebn0, args.pool, 0, dev)
            el, kms = wl.timed(25, args.config_steps, 1)
            roof, valu = roofline_of(c, cname, dtype, 0, frames, kms)
            mi, ff = wl.stats()
            entry = {"workload": f"{cname} {dtype}, {frames} frames resident in HBM, max_iters 25, Eb/N0 {ebn0} dB",
                     "value": frames * args.config_steps / el, "unit": "codewords/s", "steps": args.config_steps,
                     "ms_per_step": el / args.config_steps * 1e3, "mean_iters_returned": mi, "frame_failure_rate": ff,
                     "roofline": roof}
            if not args.no_cpu:
                secs, used, mism = wl.compare(25, min(frames, max(256, int(2048 * 8192 / c.n()) // 8)), usable_cores())
                entry["parity"] = {"frames_compared": min(frames, max(256, int(2048 * 8192 / c.n()) // 8)), "mismatches": mism}
                failed = failed or mism != 0
            configs[key] = entry
            del wl
            torch.cuda.empty_cache()
        result["configs"] = configs

    if rank == 0:
        if failed:
            result["error"] = "GPU results differ from the CPU oracle on the compared frames: the throughput is void"
            result["value"] = None
        print(json.dumps(result), flush=True)
    finish_ranks()
    return 1 if failed else 0


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
    ap.add_argument("--devices", default="", help="comma-separated device ordinal per local rank (default: rank r -> device r)")
    ap.add_argument("--cpu-frames", type=int, default=16384,
                    help="all-core cpu_baseline sample in TM8192-sized frames (one core: 1/16 of it)")
    ap.add_argument("--config-steps", type=int, default=5, help="timed launches per extra BASELINE config")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2, 3, 5")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start one rank per GPU ourselves.  Nothing above imported torch or touched a GPU,
        # and the children are fresh interpreters (no exec of a GPU-initialised process).
        # sharding.py is loaded by path: importing the package would load the HIP runtime into this process.
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ldpc_sharding", os.path.join(ROOT, "labrador_ldpc_amd", "sharding.py"))
        sharding = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharding)
        sys.exit(sharding.spawn_local_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
