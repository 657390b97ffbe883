"""f64 decode_ms: register-kernel variants vs the workspace kernel (variant 100), rates and equality.
    python tools/f64_variants.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode, LdpcHipError
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
CASES = {"TC128": (5.0, 262144), "TC256": (5.0, 262144), "TC512": (5.0, 131072), "TM1280": (4.0, 209715), "TM1536": (3.0, 174762),
         "TM2048": (3.0, 131072), "TM5120": (4.0, 52428), "TM6144": (3.0, 43690), "TM8192": (2.0, 16384)}
for name, (eb, frames) in CASES.items():
    code = LDPCCode[name]
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    f64 = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32").double()
    ref = None
    line = [f"{name} @{eb} dB {frames} frames:"]
    for variant in (100, 1, 17, 2, 18, 33, 34, 36, 0):
        try:
            out = code.decode_ms_batch(f64, 25, variant=variant); torch.cuda.synchronize()
        except LdpcHipError:
            continue
        dt = 1e9
        for _ in range(4):                       # best of four
            t = time.perf_counter(); out = code.decode_ms_batch(f64, 25, variant=variant); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t)
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        line.append(f"v{variant} {frames/dt/1e6:.3f} M/s{'' if same else ' MISMATCH'}")
    print("  ".join(line), flush=True)
