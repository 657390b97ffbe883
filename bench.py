#!/usr/bin/env python3
"""bench.py -- decoded codewords/s of the batched min-sum decoder on MI355X.

Metric (BASELINE.json): decoded codewords/sec @ 25 min-sum iterations, TM8192, Eb/N0 = 2 dB.
A "step" is one pass of the hot path (one decode_ms kernel launch through the C ABI) over this rank's
shard of synthetic AWGN frames, already resident in HBM.  The job is BASELINE config 4 -- 4 194 304 TM8192
frames (--total-frames) -- whatever the number of GPUs: one GPU decodes all of it (137 GB of LLRs fit its
288 GB), N GPUs a contiguous 1/N each with no collective ("scaling": "strong"; the reference's harness is
the same shape: one job, however many workers, perftest/src/main.rs:39-52).  --frames-per-gpu F switches to
F frames on every GPU instead ("weak").

Launching:  `python bench.py --gpus N` starts the N ranks itself (one child process per GPU, created
before anything in this process touches a GPU); under `python -m torch.distributed.run ... bench.py
--gpus N` (WORLD_SIZE set) the process is one rank.  The ranks synchronise through a gloo group
(barrier + a 2-element MAX of the timings): north_star's path has no data-path collective, so no RCCL.
torch is plumbing only: device memory, streams, events, the barrier.

Prints ONE JSON line on rank 0 with, beyond the contract's fields,
  roofline     -- algorithmic HBM bytes per launch / kernel time (HIP events on the launch stream) vs 8 TB/s
  valu_issue   -- the limiter the kernel actually runs into
  configs      -- N=1 only: BASELINE config 1 (one TC128 frame, CPU oracle beside the single-frame C entry), config 4's
                  first and last 524 288-frame slice (what each of 8 GPUs gets), configs 2, 3 and 5 (both operating points)
  cpu_baseline -- N=1 only: the CPU oracle (C port of the reference; Rust cannot be built in this image),
                  built -march=native on this host, on fixed samples of the frames the GPU decoded:
                  1 core and all cores; the same frames are compared bit for bit (parity)
  parity       -- every rank compares a fixed sample of its shard with the oracle; mismatches are summed over ranks
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
VALU_PEAK_G = 256 * 4 * 2 / 4 * 2.4   # 256 CUs x 4 SIMDs x 2 wave-instructions per 4 cycles x 2.4 GHz (nominal) = 1228.8 G/s
SHADER_CLOCK_UNDER_LOAD_GHZ = 2.30    # measured: s_memtime against the 100 MHz s_memrealtime inside the kernel (tools/kbench.hip stamps)
SLICE_FRAMES = 524288                 # config 4 over 8 GPUs: what one GPU decodes

# BASELINE.json configs beyond the headline one: (key, code, dtype, frames per GPU, Eb/N0 dB)
EXTRA_CONFIGS = [
    ("config2_TC512_f32", "TC512", "f32", 65536, 2.0),
    ("config3_TM2048_f32", "TM2048", "f32", 1048576, 2.0),
    ("config5_TM5120_i8_4dB", "TM5120", "i8", 524288, 4.0),      # per-GPU slice of 4 194 304 frames over 8 GPUs (all of them fit one GPU too: 20 GiB)
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

    def step(self, maxiters, variant=0, frames=None):
        lo, hi = frames or (0, self.frames)
        self.code.decode_ms_batch(self.llrs[lo:hi], maxiters, output=self.out[lo:hi], iters=self.iters[lo:hi], success=self.succ[lo:hi],
                                  variant=variant)

    def timed(self, maxiters, steps, warmup, variant=0, barrier=lambda: None, frames=None):
        """(wall seconds of `steps` launches, mean launch duration in ms from HIP events on the launch stream); `frames` =
        (lo, hi): only that slice of the batch."""
        import numpy as np
        import torch
        for _ in range(warmup):
            self.step(maxiters, variant, frames)
        torch.cuda.synchronize()
        # ONE pair of events around the timed launches (decode_ms_batch launches on torch's current stream, where the events
        # sit): an event between two launches is a barrier packet of its own, and for launches of well under a millisecond
        # (config 2) the bubble it opens is a tenth of the kernel -- the kernels of a stream already run in order
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        for _ in range(steps):
            self.step(maxiters, variant, frames)
        b.record()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, a.elapsed_time(b) / steps

    def stats(self):
        return float(self.iters.double().mean()), 1.0 - float(self.succ.double().mean())

    def compare(self, maxiters, sample, nthreads, lib=None, offset=0):
        """Decode `sample` frames from `offset` on the CPU oracle; (seconds, threads, mismatching frames)."""
        import numpy as np
        import oracle
        sl = slice(offset, offset + sample)
        h = self.llrs[sl].cpu().numpy()
        t1 = time.perf_counter()
        o_c, it_c, ok_c, used = oracle.decode_ms_batch(self.code, h, maxiters, nthreads, lib=lib)
        cpu_s = time.perf_counter() - t1
        o_g = self.out[sl].cpu().numpy()
        it_g = self.iters[sl].cpu().numpy().astype(np.int64)
        ok_g = self.succ[sl].cpu().numpy().astype(np.int64)
        mism = int(((it_g != it_c.astype(np.int64)) | (ok_g != ok_c.astype(np.int64)) | (o_g != o_c).any(axis=1)).sum())
        return cpu_s, used, mism


def kernel_name(code_name, dtype, variant):
    return "decode_ms_pair_kernel" if (code_name == "TM8192" and (variant & 255) in (0, 32)) else "decode_ms_kernel"


def library_build_id():
    """sha256[:16] of the shared library this process loaded -- what ties a committed profile to the code that is timed."""
    import hashlib
    import labrador_ldpc_amd as la
    with open(la.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def profile_counters(key, build_id, path=None):
    """(per-launch PMC figures of the committed profile of this kernel, None) -- or (None, why not).  The figures count for a
    bench line only if they were collected on THIS library build (profiles/hbm_traffic.json records the build it profiled);
    a kernel edit that was not re-profiled must not leave a stale figure in a driver-stamped line."""
    try:
        with open(path or os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            doc = json.load(f)
    except Exception as e:
        return None, f"no profile: {e}"
    t = doc.get(key)
    if not t or not t.get("frames"):
        return None, f"no profile of {key}"
    have = t.get("library_build") or doc.get("library_build")
    if have != build_id:
        return None, f"stale profile: counters of {key} were collected on library build {have}, this run loaded {build_id}"
    return t, None


def limiter_text(code, itemsize, mean_iters):
    visits = 2 * code.paritycheck_sum() * (mean_iters + 1)
    return (f"VALU issue + LDS, not HBM: {algorithmic_bytes_per_codeword(code, itemsize)} B of I/O per "
            f"{visits / 1e3:.0f} k edge visits ({code.paritycheck_sum()} edges x 2 passes x {mean_iters + 1:.1f} iterations); see valu_issue")


def roofline_of(code, code_name, dtype, variant, frames, kernel_ms, mean_iters, build_id):
    itemsize = 4 if dtype == "f32" else 1
    bytes_per_launch = frames * algorithmic_bytes_per_codeword(code, itemsize)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    prof, why = profile_counters(f"{code_name}_{dtype}", build_id)
    traffic = prof["hbm_bytes_per_launch"] * (frames / prof["frames"]) if prof else None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": kernel_name(code_name, dtype, variant), "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch, "limiter": limiter_text(code, itemsize, mean_iters)}
    if prof:
        roof["traffic_source"] = (f"PMC FETCH_SIZE x2 + WRITE_SIZE of a {prof['frames']}-frame launch of this library build "
                                  f"({build_id}), scaled by frames (profiles/hbm_traffic.json)")
    else:
        roof["traffic_note"] = why
    valu = None
    if prof and prof.get("valu_insts_per_launch"):
        # wave-level VALU instructions (PMC SQ_INSTS_VALU of the profiled launch, scaled to this one) against the issue peak
        ach = prof["valu_insts_per_launch"] * (frames / prof["frames"]) / (kernel_ms * 1e-3) / 1e9
        at_clock = VALU_PEAK_G * SHADER_CLOCK_UNDER_LOAD_GHZ / 2.4
        valu = {"achieved": ach, "peak": VALU_PEAK_G, "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_G,
                "peak_at_measured_clock": at_clock, "frac_at_measured_clock": ach / at_clock,
                "derived": True,
                "source": f"SQ_INSTS_VALU per frame of the profiled launch (same library build {build_id}, same code, Eb/N0 and "
                          f"iteration cap) x this run's frames / kernel time; peak at the nominal 2.4 GHz and at the "
                          f"{SHADER_CLOCK_UNDER_LOAD_GHZ} GHz the chip holds under this load"}
        if prof.get("avg_issue_cycles_per_instruction"):
            # the peak above is for 2-cycle instructions only; this kernel's mix (tools/valu_mix.py: static count over the
            # iteration loop, per-class issue cost from tools/ubench) averages more, so in VALU CYCLES the fraction is
            c = prof["avg_issue_cycles_per_instruction"]
            valu["cycle_weighted"] = {"avg_issue_cycles_per_instruction": c,
                                      "frac_of_valu_cycles_at_2.4GHz": ach / VALU_PEAK_G * c / 2.0,
                                      "frac_of_valu_cycles_at_measured_clock": ach / at_clock * c / 2.0,
                                      "note": "before the co-issue of F-class instructions with 4-cycle ones of other waves"}
    return roof, valu


def config1_leg(args):
    """BASELINE config 1: TC128, ONE codeword, 50 iterations, AWGN at 3 dB -- the reference's own CPU-runnable shape
    (capi/src/lib.rs:83-95: one frame, caller-owned buffers).  The CPU oracle and the single-frame C entry point of this
    library decode the same frame; results must be equal; both latencies are reported."""
    import numpy as np
    import oracle
    from labrador_ldpc_amd import LDPCCode
    code = LDPCCode.TC128
    rng = np.random.default_rng(0x1DBC + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 1, 3.0, np.float32)
    frame = llrs[0]
    ok_c, it_c, out_c = oracle.decode_ms(code, frame, 50)
    out_g = np.zeros(code.output_len(), dtype=np.uint8)
    ok_g, it_g = code.decode_ms(frame, out_g, maxiters=50)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.decode_ms(code, frame, 50)
    cpu_us = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        code.decode_ms(frame, out_g, maxiters=50)
    gpu_us = (time.perf_counter() - t0) / reps * 1e6
    equal = bool(ok_c == ok_g and it_c == it_g and (out_c == out_g).all())
    return {"workload": "TC128 f32, 1 codeword, max_iters 50, AWGN Eb/N0 3.0 dB (the reference's single-frame call shape)",
            "success": bool(ok_g), "iters": int(it_g), "equal_to_cpu_oracle": equal,
            "cpu_oracle_us_per_call": cpu_us, "gpu_single_frame_entry_us_per_call": gpu_us,
            "note": "latency of one call incl. the Python binding; the GPU figure is launch + two PCIe copies of one frame, "
                    "not a throughput"}, (0 if equal else 1)


def run_rank(args):
    import numpy as np
    import torch
    from labrador_ldpc_amd import LDPCCode
    from labrador_ldpc_amd.sharding import init_ranks, reduce_max, reduce_sum, barrier, finish_ranks, shard_range

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
    if args.frames_per_gpu:
        F, total, scaling = args.frames_per_gpu, args.frames_per_gpu * world, "weak"
    else:
        (_, F), total, scaling = shard_range(args.total_frames, world, rank), args.total_frames, "strong"
    build_id = library_build_id()
    w = Workload(code, args.dtype, F, args.ebn0, args.pool, rank, dev)
    elapsed, kernel_ms = w.timed(args.maxiters, args.steps, args.warmup, args.variant, barrier)
    elapsed, kernel_ms_max = reduce_max([elapsed, kernel_ms])              # slowest rank
    mean_iters, frame_fail = w.stats()

    # ---- every rank: a fixed sample of its own shard against the CPU oracle (N=1: the cpu_baseline samples below) ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    mism_rank, compared_rank = 0, 0
    if world > 1 and not args.no_cpu:
        import oracle                                  # test infrastructure: the checker
        compared_rank = min(F, args.rank_parity_frames)
        _, _, mism_rank = w.compare(args.maxiters, compared_rank, max(1, usable_cores() // world))
    if world > 1:
        mism_total, compared_total = (int(x) for x in reduce_sum([mism_rank, compared_rank]))

    result, failed = None, False
    if rank == 0:
        headline = args.code == "TM8192" and args.maxiters == 25 and args.ebn0 == 2.0 and args.dtype == "f32"
        roof, valu = roofline_of(code, args.code, args.dtype, args.variant, F, kernel_ms_max, mean_iters, build_id)
        value = total * args.steps / elapsed
        what = (f"{total} frames on {world} GPU{'s' if world > 1 else ''}" +
                (" = BASELINE config 4 whole" if headline and total == 4194304 else "") +
                (f" ({F} per GPU)" if world > 1 else ""))
        result = {
            "metric": "decoded codewords/sec @25 min-sum iters, TM8192, Eb/N0=2dB; 1/2/4/8 GPU" if headline else
                      f"decoded codewords/sec @{args.maxiters} min-sum iters, {args.code} {args.dtype}, Eb/N0={args.ebn0}dB",
            "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.code} {args.dtype} LLRs, {what}, resident in HBM, max_iters {args.maxiters} with early "
                                   f"termination, AWGN Eb/N0 {args.ebn0} dB, {args.pool} random codewords",
                       "code": args.code, "total_frames": total, "frames_per_gpu": F, "max_iters": args.maxiters, "ebn0_db": args.ebn0,
                       "sigma": w.sigma, "parallelism": f"{world} contiguous shard(s) of one batch, one process per GPU, no collective",
                       "kernel_variant": args.variant, "library_build": build_id},
            "roofline": roof, "valu_issue": valu,
            "diag": {"mean_iters_returned": mean_iters, "frame_failure_rate": frame_fail,
                     "edge_visits_per_s": value * 2 * code.paritycheck_sum() * (mean_iters + 1)},
        }
        if world > 1:
            result["parity"] = {"frames_compared": compared_total, "mismatches": mism_total,
                                "note": f"first {compared_rank} frames of every rank's shard against the CPU oracle"}
            failed = failed or mism_total != 0

    # ---- N=1 only: CPU baseline + parity on fixed samples, then the other BASELINE configs --------------
    if rank == 0 and world == 1 and not args.no_cpu:
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
        # and the far end of the batch (the last frames a one-GPU run of config 4 decodes live 137 GB into the buffer)
        tail = min(F, 2048)
        _, _, mism_tail = w.compare(args.maxiters, tail, cores, native, offset=F - tail)
        result["cpu_baseline"] = {
            "value": s_all / t_all, "unit": "codewords/s", "cores": used, "kind": "port",
            "one_core": {"value": s_one / t_one, "unit": "codewords/s", "cores": 1, "sample": f"first {s_one} frames, {t_one:.1f} s"},
            "sample": f"first {s_all} of the {F} frames the GPU decoded, same LLR bits, {t_all:.1f} s wall on {used} threads "
                      f"(C port of decode_ms, {build}; the Rust reference cannot be built in this image)"}
        result["parity"] = {"frames_compared": s_all + tail, "mismatches": mism + mism1 + mism_tail,
                            "note": f"first {s_all} and last {tail} frames of the batch against the CPU oracle"}
        failed = failed or (mism + mism1 + mism_tail) != 0
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0 and world == 1 and not args.no_configs:
        configs = {}
        if not args.no_cpu:
            configs["config1_TC128_1frame_50it_3dB"], bad = config1_leg(args)
            failed = failed or bad != 0
        # config 4 as each of 8 GPUs sees it: the first and the last 524 288-frame slice of the same buffer
        if args.code == "TM8192" and F >= 2 * SLICE_FRAMES:
            for name, lo in (("config4_slice_first", 0), ("config4_slice_last", F - SLICE_FRAMES)):
                el, kms = w.timed(args.maxiters, args.config_steps, 1, args.variant, frames=(lo, lo + SLICE_FRAMES))
                mi = float(w.iters[lo:lo + SLICE_FRAMES].double().mean())
                roof, _ = roofline_of(code, args.code, args.dtype, args.variant, SLICE_FRAMES, kms, mi, build_id)
                configs[name] = {"workload": f"frames [{lo}, {lo + SLICE_FRAMES}) of the headline batch: the share of one of 8 GPUs",
                                 "value": SLICE_FRAMES * args.config_steps / el, "unit": "codewords/s", "steps": args.config_steps,
                                 "ms_per_step": el / args.config_steps * 1e3, "mean_iters_returned": mi, "roofline": roof}
        del w
        torch.cuda.empty_cache()
        for key, cname, dtype, frames, ebn0 in EXTRA_CONFIGS:
            c = LDPCCode[cname]
            wl = Workload(c, dtype, frames, ebn0, args.pool, 0, dev)
            # enough back-to-back launches for ~100 ms of timed region after ~25 ms of warm-up: a launch of well under a
            # millisecond (config 2) is otherwise timed while the device is still coming out of idle -- its duration falls by
            # 15 % over the first few milliseconds of continuous work (profiles/r03_kbench/tc_trace.txt)
            _, probe_ms = wl.timed(25, 1, 1)
            steps = int(min(200, max(args.config_steps, 100.0 / max(probe_ms, 1e-3))))
            el, kms = wl.timed(25, steps, max(1, steps // 4))
            mi, ff = wl.stats()
            roof, valu = roofline_of(c, cname, dtype, 0, frames, kms, mi, build_id)
            entry = {"workload": f"{cname} {dtype}, {frames} frames resident in HBM, max_iters 25, Eb/N0 {ebn0} dB",
                     "value": frames * steps / el, "unit": "codewords/s", "steps": steps, "warmup": max(1, steps // 4),
                     "ms_per_step": el / steps * 1e3, "mean_iters_returned": mi, "frame_failure_rate": ff,
                     "roofline": roof}
            if valu:
                entry["valu_issue"] = {k: valu[k] for k in ("achieved", "frac", "frac_at_measured_clock")}
            if not args.no_cpu:
                sample = min(frames, max(256, int(2048 * 8192 / c.n()) // 8))
                secs, used, mism = wl.compare(25, sample, usable_cores())
                entry["parity"] = {"frames_compared": sample, "mismatches": mism}
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
    ap.add_argument("--total-frames", type=int, default=4194304,
                    help="the whole job (BASELINE config 4), split into contiguous shards over the GPUs")
    ap.add_argument("--frames-per-gpu", type=int, default=0,
                    help="weak scaling instead: this many frames on EVERY GPU (0 = a share of --total-frames)")
    ap.add_argument("--pool", type=int, default=256, help="distinct random codewords the frames cycle through")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--devices", default="", help="comma-separated device ordinal per local rank (default: rank r -> device r)")
    ap.add_argument("--cpu-frames", type=int, default=16384,
                    help="all-core cpu_baseline sample in TM8192-sized frames (one core: 1/16 of it)")
    ap.add_argument("--rank-parity-frames", type=int, default=256, help="N > 1: frames of every rank's shard compared with the oracle")
    ap.add_argument("--config-steps", type=int, default=5, help="timed launches per extra BASELINE config")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity legs")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 1, 2, 3, 5 and the config 4 slices")
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
