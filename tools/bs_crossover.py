"""At which batch does the bit-sliced i8 kernel (`variant` 64: one wave per group of codewords, ~2048 groups fill the chip) overtake the
f32-pipe i8 kernels (a workgroup per codeword group, 256-1024 fill it)?  Time per call of both, batch by batch, same frames --
the measurement behind bitslice_min_batch() in csrc/decode_ms_i8.hip.     python tools/bs_crossover.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
CASES = ((LDPCCode.TM8192, 2.0), (LDPCCode.TM2048, 2.5), (LDPCCode.TM5120, 4.0), (LDPCCode.TM6144, 3.0), (LDPCCode.TM1536, 3.0), (LDPCCode.TM1280, 4.0))
if len(sys.argv) > 1:                                           # python tools/bs_crossover.py TM5120 TM1280
    CASES = tuple(c for c in CASES if c[0].name in sys.argv[1:])
for code, ebn0 in CASES:
    G = 64 // (code.submatrix_size() // 32)
    rng = np.random.default_rng(1)
    pool = np.zeros((16, code.n() // 8), np.uint8)
    for i in range(16):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    top = 65536 if code in (LDPCCode.TM1280, LDPCCode.TM1536) else 8192
    llrs = code.awgn_frames(torch.from_numpy(pool).to(dev), top * G, sigma, seed=5, dtype="i8")
    old = 32 if code == LDPCCode.TM8192 else 1
    print(f"{code.name} (G = {G} codewords per wave), us per call: groups  f32-pipe  bit-sliced  (bit-sliced lockstep, 64 | 256)", flush=True)
    for groups in (1, 16, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192, 16384, 32768, 65536):
        if groups > top: break
        l = llrs[: groups * G]
        t = {}
        for name, variant in (("old", old), ("bs", 64), ("ls", 64 | 256)):
            for _ in range(3):
                code.decode_ms_batch(l, 25, variant=variant)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if groups <= 1024 else 5
            a.record()
            for _ in range(reps):
                code.decode_ms_batch(l, 25, variant=variant)
            b.record(); torch.cuda.synchronize()
            t[name] = a.elapsed_time(b) / reps * 1e3
        print(f"   {groups:6d} {t['old']:10.1f} {t['bs']:10.1f} {t['ls']:10.1f}   {'bit-sliced' if t['bs'] < t['old'] else 'f32-pipe'}", flush=True)
