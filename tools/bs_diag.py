"""Rate of the bit-sliced i8 kernel (variant 64) on fixed 25-iteration work (2 dB: nothing converges on the rate-4/5 code; for the
other codes the frames are pure noise), for diagnostic builds selected with LABRADOR_LDPC_HIP_LIB (tools/bs_diag_build.sh).
    python tools/bs_diag.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
CASES = ((LDPCCode.TM5120, 262144), (LDPCCode.TM1280, 1048576), (LDPCCode.TM8192, 131072), (LDPCCode.TM2048, 524288), (LDPCCode.TM6144, 131072))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if c[0].name in sys.argv[1:])
for code, frames in CASES:
    g = torch.Generator(device=dev); g.manual_seed(3)
    llrs = torch.randint(-40, 41, (frames, code.n()), dtype=torch.int8, device=dev, generator=g)      # noise: 25 iterations each
    out = code.decode_ms_batch(llrs, 25, variant=64); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): out = code.decode_ms_batch(llrs, 25, variant=64)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"{code.name}: {frames} noise frames, mean iters {float(out[1].double().mean()):.2f}: {frames / ms / 1e3:8.2f} M codewords/s = {frames * 25 / ms / 1e3:8.1f} M codeword-iterations/s", flush=True)
