"""i16 LLRs on TM8192 / TM2048: today's kernel (the f32 pipe) against the 16-plane build of the bit-sliced kernel text
(tools/bs_alt_build.sh "p16w2:-DBS_PLANES=16 -DBS_WAVES_R12=2": i16 saturation, fed sign-extended i8 LLRs through the i8 loader), same frames.
    python tools/i16_rate.py                                   -> rates of decode_ms_batch on int16 tensors (default library)
    LABRADOR_LDPC_HIP_LIB=build/alt/liblabrador_ldpc_hip_p16w2.so python tools/i16_rate.py bs16    -> the 16-plane bit-sliced kernel
Both print a digest of (output, iters, success): the two must agree (the frames' LLRs lie in the i8 range)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
mode = sys.argv[1] if len(sys.argv) > 1 else "f32pipe"
dev = torch.device("cuda", 0)
print("library:", os.environ.get("LABRADOR_LDPC_HIP_LIB", "(default)"), "mode", mode, flush=True)
for name, ebn0, frames in (("TM8192", 2.0, 262144), ("TM2048", 2.0, 1048576), ("TM2048", 2.5, 1048576)):
    code = LDPCCode[name]
    rng = np.random.default_rng(1)
    pool = np.zeros((64, code.n() // 8), np.uint8)
    for i in range(64):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    l8 = code.awgn_frames(torch.from_numpy(pool).to(dev), frames, sigma, seed=5, dtype="i8")
    if mode == "bs16":
        llrs, kw = l8, dict(variant=64)
    else:
        llrs, kw = l8.to(torch.int16), {}
    out = code.decode_ms_batch(llrs, 25, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            out = code.decode_ms_batch(llrs, 25, **kw)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 3)
    h = hashlib.sha256()
    for t in out:
        h.update(t.cpu().numpy().tobytes())
    print(f"{name} {ebn0} dB {frames} frames  {frames / best / 1e3:8.2f} M codewords/s  {best:8.2f} ms  mean iters {float(out[1].double().mean()):.2f}  digest {h.hexdigest()[:16]}", flush=True)
