"""Rate-4/5 codes, i8 LLRs: the two-waves-per-group bit-sliced kernel (`variant` 128) against the one-wave kernel (`variant` 64) and the
f32-pipe i8 kernel, same frames, same process; outputs compared.     python tools/split_rate.py [frames of TM5120]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 419430
dev = torch.device("cuda", 0)
for code, ebn0 in ((LDPCCode.TM5120, 4.0), (LDPCCode.TM5120, 2.0), (LDPCCode.TM1280, 4.0), (LDPCCode.TM1280, 2.0)):
    rng = np.random.default_rng(1)
    pool = np.zeros((64, code.n() // 8), np.uint8)
    for i in range(64):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    fr = frames * 5120 // code.n()
    llrs8 = code.awgn_frames(torch.from_numpy(pool).to(dev), fr, sigma, seed=5, dtype="i8")
    res = {}
    for name, variant in (("f32-pipe", 1), ("bit-sliced, one wave", 128), ("bit-sliced, two waves", 64)):
        out = code.decode_ms_batch(llrs8, 25, variant=variant)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            out = code.decode_ms_batch(llrs8, 25, variant=variant)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        res[name] = out
        print(f"{code.name} {ebn0} dB {fr} frames  {name:22s} {fr / ms / 1e3:8.2f} M codewords/s  {ms:8.2f} ms  mean iters {float(out[1].double().mean()):.2f}", flush=True)
    print("   all three equal:", all(torch.equal(x, y) for k in ("bit-sliced, one wave", "bit-sliced, two waves") for x, y in zip(res["f32-pipe"], res[k])), flush=True)
