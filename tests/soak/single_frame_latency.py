"""Latency of the reference-shaped single-frame entry points (labrador_ldpc_decode_ms_*, capi/src/lib.rs:83-95) through the
library: one frame, host pointers.  Run twice to compare the direct small-call path with the copying one:
    python tests/soak/single_frame_latency.py; LABRADOR_LDPC_HIP_NO_DIRECT=1 python tests/soak/single_frame_latency.py   (under tests/: it checks against the oracle)"""
import ctypes, os, sys, time, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import labrador_ldpc_amd as la
from labrador_ldpc_amd import LDPCCode
import oracle

mode = "copies (LABRADOR_LDPC_HIP_NO_DIRECT)" if os.environ.get("LABRADOR_LDPC_HIP_NO_DIRECT") else \
       "direct, hipStreamSynchronize (LABRADOR_LDPC_HIP_NO_NOTIFY)" if os.environ.get("LABRADOR_LDPC_HIP_NO_NOTIFY") else "direct, completion ticket"
print(f"single-frame calls, {mode}: us per call (median of 5 x 400 calls), results checked against the CPU oracle")
rng = np.random.default_rng(5)
for code, ebn0 in ((LDPCCode.TC128, 3.0), (LDPCCode.TC512, 3.0), (LDPCCode.TM1280, 4.0), (LDPCCode.TM2048, 2.5), (LDPCCode.TM5120, 4.0), (LDPCCode.TM8192, 2.0)):
    row = []
    for dtype, fn in ((np.float32, la.lib.labrador_ldpc_decode_ms_f32), (np.int8, la.lib.labrador_ldpc_decode_ms_i8)):
        llrs, _ = oracle.awgn_llrs(code, rng, 1, ebn0, dtype)
        out = np.zeros(code.output_len(), dtype=np.uint8)
        it = ctypes.c_size_t(0)
        args = (int(code), llrs.ctypes.data, out.ctypes.data, None, None, 50, ctypes.byref(it))
        try:
            ok = fn(*args)
        except Exception as e:
            row.append(f"{np.dtype(dtype).name}: n/a ({e})"); continue
        o_c, i_c, k_c, _ = oracle.decode_ms_batch(code, llrs, 50)
        assert bool(ok) == bool(k_c[0]) and (out == o_c[0]).all() and (not ok or it.value == i_c[0]), (code, dtype)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(400):
                fn(*args)
            ts.append((time.perf_counter() - t) / 400 * 1e6)
        row.append(f"{np.dtype(dtype).name} {sorted(ts)[2]:6.1f}")
    print(f"{code.name:7s} " + "   ".join(row), flush=True)
