"""Rate of the bit-sliced i8 kernels (`variant` 64) alone, with a digest of every output, for A/B runs of alternative library builds:
    LABRADOR_LDPC_HIP_LIB=build/diag/liblabrador_ldpc_hip_<name>.so python tools/bs_ab.py [frames-of-TM8192-size] [case ...]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
want = [a for a in sys.argv[2:] if not a.startswith("variant=")]
VARIANT = ([int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("variant=")] or [64])[0]   # 64 | 256 = the lockstep kernels (no slot refill)
dev = torch.device("cuda", 0)
CASES = (("TM8192", 2.0), ("TM2048", 2.0), ("TM2048", 2.5), ("TM6144", 3.0), ("TM1536", 3.0), ("TM5120", 4.0), ("TM5120", 2.0), ("TM1280", 4.0))
print("library:", os.environ.get("LABRADOR_LDPC_HIP_LIB", "(default)"), "variant", VARIANT, flush=True)
for name, ebn0 in CASES:
    if want and name not in want:
        continue
    code = LDPCCode[name]
    rng = np.random.default_rng(1)
    pool = np.zeros((64, code.n() // 8), np.uint8)
    for i in range(64):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    fr = frames * 8192 // code.n()
    llrs8 = code.awgn_frames(torch.from_numpy(pool).to(dev), fr, sigma, seed=5, dtype="i8")
    out = code.decode_ms_batch(llrs8, 25, variant=VARIANT)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            out = code.decode_ms_batch(llrs8, 25, variant=VARIANT)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 3)
    h = hashlib.sha256()
    for t in out:
        h.update(t.cpu().numpy().tobytes())
    print(f"{name} {ebn0} dB {fr} frames  {fr / best / 1e3:8.2f} M codewords/s  {best:8.2f} ms  mean iters {float(out[1].double().mean()):.2f}  digest {h.hexdigest()[:16]}", flush=True)
