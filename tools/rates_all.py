"""Library-level decode rates (device-resident frames) for every code and LLR type at a fixed operating point.
    python tools/rates_all.py > gpurun_out/rates_all.txt"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
EBN0 = {"TC128": 5.0, "TC256": 5.0, "TC512": 5.0, "TM1280": 4.0, "TM1536": 3.0, "TM2048": 2.5, "TM5120": 4.0, "TM6144": 3.0, "TM8192": 2.0}
def rate(fn, n):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return n / best / 1e6
print("code    Eb/N0  frames   mean_it   f32 M/s   i8 M/s  i16 M/s  i32 M/s  f64 M/s  encode M/s  decode_bf M/s")
for code in LDPCCode:
    eb = EBN0[code.name]
    frames = max(16384, min(1048576, (1 << 31) // (code.n() * 8)))
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    f32 = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32")
    i8 = code.awgn_frames(cws, frames, sigma, seed=5, dtype="i8")
    i16 = (f32 * 64).round().clamp(-32000, 32000).to(torch.int16)
    _, it, _ = code.decode_ms_batch(f32, 25)
    r32 = rate(lambda: code.decode_ms_batch(f32, 25), frames)
    r8 = rate(lambda: code.decode_ms_batch(i8, 25), frames)
    r16 = rate(lambda: code.decode_ms_batch(i16, 25), frames)
    i32 = (f32.double() * 1e6).round().clamp(-2e9, 2e9).to(torch.int32)
    ri32 = rate(lambda: code.decode_ms_batch(i32, 25), frames)
    del i32
    f64frames = frames if code.name != "TM8192" else 16384
    f64 = f32[:f64frames].double()
    r64 = rate(lambda: code.decode_ms_batch(f64, 25), f64frames)
    del f64, i16
    d = torch.randint(0, 256, (frames, code.k() // 8), dtype=torch.uint8, device=dev)
    renc = rate(lambda: code.encode_batch(d), frames)
    hard = cws[torch.arange(frames, device=dev) % 256].clone()
    idx = torch.randint(0, code.n() // 8, (frames,), device=dev)
    hard[torch.arange(frames, device=dev), idx] ^= 0x10
    rbf = rate(lambda: code.decode_bf_batch(hard, 50), frames)
    print(f"{code.name:7s} {eb:4.1f} {frames:8d} {float(it.float().mean()):8.2f} {r32:9.2f} {r8:8.2f} {r16:8.2f} {ri32:8.2f} {r64:8.3f} {renc:10.1f} {rbf:10.1f}", flush=True)
