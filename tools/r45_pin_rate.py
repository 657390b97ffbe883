import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
for code, ebn0 in ((LDPCCode.TM5120, 4.0), (LDPCCode.TM5120, 2.0), (LDPCCode.TM1280, 4.0)):
    rng = np.random.default_rng(1)
    pool = np.zeros((64, code.n() // 8), np.uint8)
    for i in range(64):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * 0.8 * 10.0 ** (ebn0 / 10.0))))
    fr = 419430 if code == LDPCCode.TM5120 else 1677721
    llrs = code.awgn_frames(torch.from_numpy(pool).to(dev), fr, sigma, seed=5, dtype="i8")
    for _ in range(2): out = code.decode_ms_batch(llrs, 25, variant=64)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): out = code.decode_ms_batch(llrs, 25, variant=64)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(os.environ.get("LABRADOR_LDPC_HIP_LIB", "default"), code.name, ebn0, f"{fr / ms / 1e3:.2f} M cw/s", f"sum iters {int(out[1].sum())}", flush=True)
