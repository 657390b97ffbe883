"""Rate of the bit-sliced i8 kernel (`variant` 64) against the default i8 kernel and the f32 kernel, same frames, same process.
    python tools/bs_rate.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda", 0)
CASES = ((LDPCCode.TM8192, 2.0), (LDPCCode.TM2048, 2.0), (LDPCCode.TM2048, 2.5), (LDPCCode.TM6144, 3.0), (LDPCCode.TM1536, 3.0),
         (LDPCCode.TM5120, 4.0), (LDPCCode.TM5120, 2.0), (LDPCCode.TM1280, 4.0))
for code, ebn0 in CASES:
    rng = np.random.default_rng(1)
    pool = np.zeros((64, code.n() // 8), np.uint8)
    for i in range(64):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    fr = frames * 8192 // code.n()
    llrs8 = code.awgn_frames(torch.from_numpy(pool).to(dev), fr, sigma, seed=5, dtype="i8")
    llrs32 = code.awgn_frames(torch.from_numpy(pool).to(dev), fr, sigma, seed=5, dtype="f32")
    res = {}
    old = 32 if code == LDPCCode.TM8192 else 1           # the f32-pipe i8 kernels by their explicit variant (the default is bit-sliced now)
    for name, l, variant in (("i8 f32-pipe", llrs8, old), ("i8 bit-sliced", llrs8, 64), ("f32 default", llrs32, 0)):
        out = code.decode_ms_batch(l, 25, variant=variant)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            out = code.decode_ms_batch(l, 25, variant=variant)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        res[name] = out
        print(f"{code.name} {ebn0} dB {fr} frames  {name:14s} {fr / ms / 1e3:8.2f} M codewords/s  {ms:8.2f} ms  mean iters {float(out[1].double().mean()):.2f}", flush=True)
    same = all(torch.equal(x, y) for x, y in zip(res["i8 f32-pipe"], res["i8 bit-sliced"]))
    print("   bit-sliced == default i8:", same, flush=True)
