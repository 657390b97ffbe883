import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
def rate(fn, n, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return n / best
rng = np.random.default_rng(3)
for code, eb, frames in ((LDPCCode.TC512, 5.0, 262144), (LDPCCode.TM2048, 3.0, 65536), (LDPCCode.TM8192, 2.0, 16384)):
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    f32 = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32")
    f64 = f32.double()
    i16 = (f32 * 64).round().clamp(-32000, 32000).to(torch.int16)
    r32 = rate(lambda: code.decode_ms_batch(f32, 25), frames)
    r64 = rate(lambda: code.decode_ms_batch(f64, 25), frames)
    r16 = rate(lambda: code.decode_ms_batch(i16, 25), frames)
    hard = (cws[torch.arange(frames, device=dev) % 256]).clone()
    idx = torch.randint(0, code.n() // 8, (frames,), device=dev)
    hard[torch.arange(frames, device=dev), idx] ^= 0x10
    rbf = rate(lambda: code.decode_bf_batch(hard, 50), frames)
    print(f"{code.name} @{eb} dB: f32 {r32/1e6:.2f} M/s, i16 {r16/1e6:.2f} M/s, f64 {r64/1e6:.3f} M/s, decode_bf (1 flipped bit) {rbf/1e6:.2f} M/s", flush=True)
