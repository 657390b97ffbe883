"""Pair-ownership kernel (variant 32) vs the default kernel: equality and rate.  python tools/pair_check.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for name, eb, frames, dtypes in (("TM8192", 2.0, 131072, ("f32", "i8")), ("TM2048", 2.5, 262144, ("f32",))):
    code = LDPCCode[name]
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    for dt in dtypes:
        llrs = code.awgn_frames(cws, frames, sigma, seed=5, dtype=dt)
        res = {}
        for variant in (0, 32):
            try:
                out = code.decode_ms_batch(llrs, 25, variant=variant); torch.cuda.synchronize()
            except Exception as e:
                print(name, dt, "variant", variant, "->", e); continue
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); out = code.decode_ms_batch(llrs, 25, variant=variant); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            res[variant] = (out, frames / best / 1e6)
        if 0 in res and 32 in res:
            same = all(torch.equal(a, b) for a, b in zip(res[0][0], res[32][0]))
            print(f"{name} {dt} @{eb} dB: default {res[0][1]:.3f} M/s, pair {res[32][1]:.3f} M/s, identical: {same}", flush=True)
