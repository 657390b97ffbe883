import sys, time, torch
sys.path.insert(0, '.')
from labrador_ldpc_amd import LDPCCode
dev = torch.device('cuda', 0)
for name, B in (("TC512", 1 << 20), ("TM2048", 1 << 20), ("TM8192", 1 << 19), ("TM5120", 1 << 19)):
    code = LDPCCode[name]
    data = torch.randint(0, 256, (B, code.k() // 8), dtype=torch.uint8, device=dev)
    cw = torch.empty((B, code.n() // 8), dtype=torch.uint8, device=dev)
    code.encode_batch(data, cw); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): code.encode_batch(data, cw)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"{name}: {B} frames in {ms:.2f} ms -> {B / ms / 1e3:.1f} M codewords/s, {B * (code.k() + code.n()) / 8 / ms / 1e6:.1f} GB/s of data+codeword bytes")
