import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from labrador_ldpc_amd import LDPCCode
rng = np.random.default_rng(2)
for code in (LDPCCode.TC128, LDPCCode.TC512, LDPCCode.TM2048, LDPCCode.TM8192):
    llrs, _ = oracle.awgn_llrs(code, rng, 64, 3.0, np.float32)
    out = np.zeros(code.output_len(), dtype=np.uint8)
    code.decode_ms(llrs[0], out, maxiters=25)
    t = time.perf_counter()
    for f in range(64): code.decode_ms(llrs[f], out, maxiters=25)
    dt = (time.perf_counter() - t) / 64
    t = time.perf_counter(); oracle.decode_ms(code, llrs[0], 25); dc = time.perf_counter() - t
    print(f"{code.name}: single-frame decode_ms_f32 via C ABI {dt*1e6:.0f} us/frame (CPU oracle {dc*1e6:.0f} us)")
