"""GPU encoder against the oracle, every frame, bit for bit: nine codes x batch sizes that exercise the single-workgroup, ragged,
resident-grid (XCD-aware map on) and multi-round launches of csrc/encode.hip.  Lives under tests/ because it runs the oracle.
    python tests/soak/enc_soak.py [multiplier]        (A/B switches: LABRADOR_LDPC_HIP_ENC_1COL=1, LABRADOR_LDPC_HIP_ENC_PLAIN_MAP=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from labrador_ldpc_amd import LDPCCode

mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(20261004)
total = bad = 0
t0 = time.time()
switches = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("LABRADOR_LDPC_HIP_ENC"))
for code in LDPCCode:
    kb = code.k() // 8
    big = max(4096, (mult * 6 * 1024 * 1024) // code.n())
    for B in (1, 7, 8, 9, 63, 257, 1000, 4091, 4096, 4104, big + 3):
        data = rng.integers(0, 256, (B, kb), dtype=np.uint8)
        data[rng.integers(0, B)] = 0
        data[rng.integers(0, B)] = 0xFF
        if B > 8:                                                # one set data bit: the parity is one generator row
            data[3, :] = 0
            data[3, rng.integers(0, kb)] = 1 << rng.integers(0, 8)
        cw = code.encode_batch(data)
        m = 0
        for f in range(B):
            m += int((cw[f] != oracle.copy_encode(code, data[f])).any())
        total += B; bad += m
    print(f"encode {code.name}: batches up to {big + 3} frames, running total {total}, mismatches {bad}", flush=True)
print(f"TOTAL {total} frames compared, {bad} mismatches, {time.time() - t0:.0f} s  [{switches or 'default kernels'}]")
sys.exit(1 if bad else 0)
