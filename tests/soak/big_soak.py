"""Wide parity soak: every code x LLR type, several operating points, GPU vs the oracle.
    python tests/soak/big_soak.py [multiplier]      (1 = 255 000 frames, ~75 s on the GPU box, almost all of it the CPU oracle)"""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from labrador_ldpc_amd import LDPCCode
MULT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t0 = time.time(); total = 0; bad_total = 0
for code in LDPCCode:
    frames = (1500 if code.n() <= 2048 else 400) * MULT
    for dtype in (np.float32, np.int8, np.int16, np.int32, np.float64):
        rng = np.random.default_rng(4242 + 17 * int(code) + np.dtype(dtype).itemsize + 1000 * (MULT - 1))
        for ebn0, mi in ((0.5, 8), (1.5, 30), (2.5, 25), (4.0, 25), (7.0, 10)):
            scale, lim = (8.0, 31) if dtype == np.int8 else ((3e8, 2 ** 31 - 1) if dtype == np.int32 else (64.0, 4095))
            llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, dtype, scale=scale, lim=lim)
            o, it, ok = code.decode_ms_batch(llrs, mi)
            oc, itc, okc, _ = oracle.decode_ms_batch(code, llrs, mi)
            bad = int(((it != itc) | (ok != okc) | (o != oc).any(axis=1)).sum())
            total += frames; bad_total += bad
            if bad: print("MISMATCH", code.name, np.dtype(dtype).name, ebn0, bad, flush=True)
    print(code.name, "done", f"{time.time()-t0:.0f}s", flush=True)
print("frames compared", total, "mismatching", bad_total)
