"""Rates of the batched LLR conversions (csrc/llr_convert.hip) on device-resident buffers, against the HBM roofline:
hard_to_llrs writes 8 * sizeof(T) bytes per byte read, llrs_to_hard reads them.
    python tools/llr_rate.py > gpurun_out/llr_convert_rates.txt"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode

dev = torch.device("cuda", 0)
code = LDPCCode.TM8192
SZ = {"i8": 1, "i16": 2, "i32": 4, "f32": 4, "f64": 8}
print("code   type   frames   hard_to_llrs: ms   GB/s  % of 8 TB/s | llrs_to_hard: ms   GB/s  % of 8 TB/s")
for dt, frames in (("i8", 1048576), ("i16", 524288), ("f32", 262144), ("i32", 262144), ("f64", 131072)):
    bits = torch.randint(0, 256, (frames, code.n() // 8), dtype=torch.uint8, device=dev)
    llrs = code.hard_to_llrs_batch(bits, dt)
    out = code.llrs_to_hard_batch(llrs)
    assert torch.equal(out, bits)
    nbytes = frames * (code.n() // 8) * (1 + 8 * SZ[dt])
    res = []
    for fn in (lambda: code.hard_to_llrs_batch(bits, dt, llrs=llrs), lambda: code.llrs_to_hard_batch(llrs, output=out)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append((ms, nbytes / ms / 1e6))
    print(f"{code.name} {dt:>5s} {frames:8d}   {res[0][0]:14.3f} {res[0][1]:6.0f} {res[0][1] / 80:6.1f}       | {res[1][0]:12.3f} {res[1][1]:6.0f} {res[1][1] / 80:6.1f}")
