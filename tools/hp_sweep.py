"""Chunk-size sweep of the host-pointer pipeline (LABRADOR_LDPC_HIP_CHUNK), outputs preallocated.
    python tools/hp_sweep.py [CODE dtype frames ebn0 chunk,chunk,...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode

def sweep(code, dtype, frames, eb, chunks):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    llrs_d = code.awgn_frames(cws, frames, sigma, seed=99, dtype=dtype)
    llrs_h = llrs_d.cpu().numpy(); del llrs_d
    out = np.zeros((frames, code.output_len()), np.uint8); it = np.zeros(frames, np.uint32); ok = np.zeros(frames, np.uint8)
    for chunk in chunks:
        os.environ["LABRADOR_LDPC_HIP_CHUNK"] = str(chunk)
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); code.decode_ms_batch(llrs_h, 25, output=out, iters=it, success=ok); best = min(best, time.perf_counter() - t)
        print(f"{code.name} {dtype} chunk {chunk:7d}: {best*1e3:7.1f} ms  {frames/best/1e6:.3f} M/s  {llrs_h.nbytes/best/1e9:.1f} GB/s", flush=True)
    d = torch.empty(llrs_h.shape, dtype=torch.from_numpy(llrs_h).dtype, device=dev)
    t = time.perf_counter(); d.copy_(torch.from_numpy(llrs_h)); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{code.name} {dtype} pageable copy alone: {dt*1e3:.1f} ms {llrs_h.nbytes/dt/1e9:.1f} GB/s")

if __name__ == "__main__":
    if len(sys.argv) > 1:
        sweep(LDPCCode[sys.argv[1]], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), [int(x) for x in sys.argv[5].split(",")])
    else:
        sweep(LDPCCode.TM8192, "f32", 131072, 2.0, (2048, 8192, 32768, 131072))
        sweep(LDPCCode.TM8192, "i8", 131072, 3.0, (2048, 8192, 32768, 131072))
