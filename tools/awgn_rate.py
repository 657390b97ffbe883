"""Rate of the on-device AWGN frame generator (labrador_ldpc_hip_awgn_*; Philox + Box-Muller, csrc/channel.hip).
    python tools/awgn_rate.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
for name, dt, frames in (("TM8192", "f32", 131072), ("TM8192", "i8", 524288), ("TC512", "f32", 2097152)):
    code = LDPCCode[name]
    pool = torch.randint(0, 256, (64, code.n() // 8), dtype=torch.uint8, device=dev)
    out = code.awgn_frames(pool, frames, 0.8, seed=1, dtype=dt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(5):
        code.awgn_frames(pool, frames, 0.8, seed=2 + r, dtype=dt, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nbytes = out.numel() * out.element_size()
    print(f"{name} {dt} {frames} frames: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, {nbytes / ms / 1e6:.0f} GB/s written, {out.numel() / ms / 1e6:.1f} G samples/s")
