"""TM8192 (and TM2048) at the metric's operating point with the LLR types the f32 / bit-sliced kernels do not cover: i16, i32, f64 beside
f32 -- rates, passes executed, and (under rocprofv3 --pmc, tools/r06_wide.sh) the instruction counters behind the fraction-of-ceiling
figures of profiles/r06_final/rates_all_codes.txt.
    python tools/wide_types.py [frames]"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
def rate(fn, n, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return n / best / 1e6, best * 1e3
for name in (sys.argv[2:] or ["TM8192"]):
    code = LDPCCode[name]
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (2.0 / 10.0))))
    f32 = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32")
    llrs = {"f32": f32, "i16": (f32 * 64).round().clamp(-32000, 32000).to(torch.int16),
            "i32": (f32.double() * 1e6).round().clamp(-2e9, 2e9).to(torch.int32), "f64": f32.double()}
    for t, x in llrs.items():
        _, it, ok = code.decode_ms_batch(x, 25)
        passes = float((it.double() + ok.double()).mean())
        r, ms = rate(lambda: code.decode_ms_batch(x, 25), frames)
        print(f"{name} {t:4s} {frames} frames  {r:8.3f} M codewords/s  {ms:8.3f} ms  passes executed {passes:.3f}  edge-passes/s {r * 1e6 * code.paritycheck_sum() * passes / 1e12:.3f} T", flush=True)
