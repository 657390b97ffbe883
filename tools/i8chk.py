import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
code = LDPCCode.TM8192; dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
cws = code.encode_batch(torch.from_numpy(data).to(dev))
for eb in (2.0, 3.0):
    sigma = float(np.sqrt(1.0 / (2.0 * 0.5 * 10.0 ** (eb / 10.0))))
    i8 = code.awgn_frames(cws, 131072, sigma, seed=5, dtype="i8")
    out = torch.empty((131072, code.output_len()), dtype=torch.uint8, device=dev); it = torch.empty(131072, dtype=torch.int32, device=dev); ok = torch.empty(131072, dtype=torch.uint8, device=dev)
    for v in (0, 2, 32, 0, 2):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter(); code.decode_ms_batch(i8, 25, output=out, iters=it, success=ok, variant=v); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        print(f"i8 {eb} dB variant {v}: {131072 / best / 1e6:.3f} M/s", flush=True)
