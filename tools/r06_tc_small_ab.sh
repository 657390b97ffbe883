#!/bin/bash
# TC128 / TC256 f32 with their LLRs in LDS at four waves per SIMD (build/alt/liblabrador_ldpc_hip_tcall.so: llr_in_lds() for CODE <= TC512) against
# the in-tree library (TC512 only), same box.
mkdir -p gpurun_out/tc
cat > /tmp/tc.py <<'P'
import sys, os, time, hashlib, numpy as np, torch
sys.path.insert(0, os.getcwd())
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
for name in ("TC128", "TC256"):
    code = LDPCCode[name]
    rng = np.random.default_rng(1)
    pool = np.zeros((256, code.n() // 8), np.uint8)
    for i in range(256):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    for ebn0, frames in ((2.0, 1048576), (3.0, 1048576), (5.0, 1048576)):
        sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
        llrs = code.awgn_frames(torch.from_numpy(pool).to(dev), frames, sigma, seed=5, dtype="f32")
        out = code.decode_ms_batch(llrs, 25); torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): out = code.decode_ms_batch(llrs, 25)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 10)
        h = hashlib.sha256()
        for t in out: h.update(t.cpu().numpy().tobytes())
        print(f"{code.name} f32 {ebn0} dB {frames} frames {frames / best / 1e3:8.2f} M codewords/s  mean iters {float(out[1].double().mean()):.2f} digest {h.hexdigest()[:12]}", flush=True)
P
for l in new alt new alt; do if [ $l = alt ]; then export LABRADOR_LDPC_HIP_LIB=$PWD/build/alt/liblabrador_ldpc_hip_tcall.so; else unset LABRADOR_LDPC_HIP_LIB; fi; echo "== $l"; python3 /tmp/tc.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/tc/tc128_256_ab.txt
