"""Rate of decode_bf_batch on device-resident frames for the TM codes (a few random bit errors per frame).  The bit-sliced kernel is the
default from 256 groups up; run with LABRADOR_LDPC_HIP_BF_BYTES=1 for the byte-per-variable kernel.    python tools/bf_rate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
which = "byte-per-variable" if os.environ.get("LABRADOR_LDPC_HIP_BF_BYTES") else "bit-sliced"
for code in [c for c in LDPCCode if c.name.startswith("TM")]:
    rng = np.random.default_rng(1)
    frames = 262144 * 8192 // code.n()
    pool = np.zeros((256, code.n() // 8), np.uint8)
    for i in range(256):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
        for pos in rng.choice(code.n(), int(rng.integers(0, code.n() // 100)), replace=False):
            pool[i, pos // 8] ^= 1 << (7 - pos % 8)
    hard = torch.from_numpy(pool).to(dev)[torch.arange(frames, device=dev) % 256].contiguous()
    out = code.decode_bf_batch(hard, 50)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = code.decode_bf_batch(hard, 50)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    io = code.n() // 8 + code.output_len() + 5
    print(f"{code.name} {frames} frames  {which:18s} {frames / ms / 1e3:9.1f} M codewords/s  {ms:7.3f} ms  {frames * io / ms / 1e6:7.1f} GB/s of I/O  "
          f"mean iters {float(out[1].double().mean()):.2f} success {float(out[2].double().mean()):.4f}", flush=True)
