"""Every built kernel variant of every code (f32, i8, i16) against the tuned default: rates and equality.
    python tools/variant_sweep.py > gpurun_out/variant_sweep.txt"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode, LdpcHipError
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
EBN0 = {"TC128": 5.0, "TC256": 5.0, "TC512": 5.0, "TM1280": 4.0, "TM1536": 3.0, "TM2048": 2.5, "TM5120": 4.0, "TM6144": 3.0, "TM8192": 2.0}
for code in LDPCCode:
    eb = EBN0[code.name]
    frames = max(16384, min(1048576, (1 << 31) // (code.n() * 8)))
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    for dt in ("f32", "i8", "i16"):
        llrs = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32" if dt == "i16" else dt)
        if dt == "i16":
            llrs = (llrs * 64).round().clamp(-32000, 32000).to(torch.int16)
        ref, line = None, [f"{code.name:7s} {dt:3s} {frames:8d} frames:"]
        for variant in (0, 1, 2, 4, 32, 256, 32 + 256):
            try:
                out = code.decode_ms_batch(llrs, 25, variant=variant); torch.cuda.synchronize()
            except LdpcHipError:
                continue
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); out = code.decode_ms_batch(llrs, 25, variant=variant); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            if ref is None:
                ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            line.append(f"v{variant} {frames / best / 1e6:.2f}{'' if same else ' MISMATCH'}")
        print("  ".join(line), flush=True)
        del llrs
