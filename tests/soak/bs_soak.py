"""Wide GPU-vs-oracle parity soak of the bit-sliced kernels (forced with `variant` 64 whatever the batch; decode_bf through
its default dispatch on batches above its threshold): six TM codes x four operating points of decode_ms i8 (converging early / late /
failing / saturating scale) and two error densities of decode_bf.  Lives under tests/ because it runs the oracle.
    python tests/soak/bs_soak.py [multiplier]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from labrador_ldpc_amd import LDPCCode

mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(20261003)
total = bad = 0
t0 = time.time()
for code in [c for c in LDPCCode if c.name.startswith("TM")]:
    G = 64 // (code.submatrix_size() // 32)
    r45, r23 = code.k() * 5 == code.n() * 4, code.k() * 3 == code.n() * 2
    work = 4.0 if r45 else (3.0 if r23 else 2.2)
    frames = max(3 * G + 1, (1536 * mult * 2048) // code.n())
    for ebn0, scale, lim in ((work, 8.0, 31), (work + 1.5, 8.0, 31), (work - 1.5, 8.0, 31), (work, 40.0, 127)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, np.int8, scale=scale, lim=lim)
        o, i, k = code.decode_ms_batch(llrs, 25, variant=64)
        oc, ic, kc, _ = oracle.decode_ms_batch(code, llrs, 25)
        m = int(((o != oc).any(axis=1) | (i != ic) | (k != kc)).sum())
        total += frames; bad += m
        print(f"decode_ms i8 bit-sliced {code.name} Eb/N0 {ebn0:.1f} scale {scale:g}: {frames} frames, mean iters {i.mean():.2f}, failed {1 - k.mean():.3f}, mismatches {m}", flush=True)
    fb = max(256 * G + 5, frames)
    for dens in (400, 60):
        hard = np.zeros((fb, code.n() // 8), np.uint8)
        pool = [oracle.copy_encode(code, rng.integers(0, 256, code.k() // 8, dtype=np.uint8)) for _ in range(16)]
        for f in range(fb):
            cw = pool[f % 16].copy()
            for pos in rng.choice(code.n(), int(rng.integers(0, max(2, code.n() // dens))), replace=False):
                cw[pos // 8] ^= 1 << (7 - pos % 8)
            hard[f] = cw
        o, i, k = code.decode_bf_batch(hard, 30)
        m = 0
        for f in range(0, fb, max(1, fb // (600 * mult))):                     # the oracle's decode_bf is one frame per call: a sample
            ok_c, it_c, out_c = oracle.decode_bf(code, hard[f], 30)
            m += int(not ((bool(k[f]), int(i[f])) == (ok_c, it_c) and (o[f] == out_c).all()))
            total += 1
        bad += m
        print(f"decode_bf bit-sliced {code.name} <= n/{dens} errors: {fb} frames (sampled), success {k.mean():.3f}, mismatches {m}", flush=True)
print(f"TOTAL {total} frames compared, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
