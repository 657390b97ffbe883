"""Randomised soak of the slot-refill kernels (TM1536: decode_refill, TM1280: decode_refill_split): random batch sizes around the chunk
and grid sizes, random iteration caps and operating points, frames of mixed convergence shuffled, launches back to back on one stream
and on a second stream.  Every launch is compared with the lockstep kernel's results on the same frames, a sample of every launch with
the oracle.  Lives under tests/ because it runs the oracle.     python tests/soak/refill_fuzz.py [trials]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import oracle
from labrador_ldpc_amd import LDPCCode

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(20261004)
dev = torch.device("cuda", 0)
pools = {}
for code in (LDPCCode.TM1536, LDPCCode.TM1280):
    hi = 1.5 if code is LDPCCode.TM1280 else 0.0
    parts = [oracle.awgn_llrs(code, rng, 1024, e, np.int8, scale=s, lim=l)[0] for e, s, l in
             ((2.5 + hi, 8.0, 31), (1.0, 30.0, 127), (4.5 + hi, 8.0, 31), (3.0 + hi, 16.0, 63), (2.0 + hi, 8.0, 31))]
    pools[code] = np.concatenate(parts)
side = torch.cuda.Stream()
total = bad = checked = 0
t0 = time.time()
for t in range(trials):
    code = (LDPCCode.TM1536, LDPCCode.TM1280)[t % 2]
    pool = pools[code]
    G = 64 // (code.submatrix_size() // 32)
    kind = int(rng.integers(0, 5))
    n = int({0: rng.integers(1, 4 * G + 2), 1: rng.integers(1, 2000), 2: rng.integers(2000, 40000), 3: rng.integers(40000, 140000),
             4: 16384 + rng.integers(-2, 3) * G + rng.integers(-1, 2)}[kind])
    maxiters = int(rng.choice([0, 1, 2, 3, 7, 25, 25, 25, 40]))
    idx = rng.integers(0, pool.shape[0], n)
    llrs = torch.from_numpy(pool[idx]).to(dev)
    ref = code.decode_ms_batch(llrs, maxiters, variant=64 | 256)
    if t % 3 == 2:
        with torch.cuda.stream(side):
            got = code.decode_ms_batch(llrs, maxiters, variant=64)
            again = code.decode_ms_batch(llrs, maxiters, variant=64)
        side.synchronize()
    else:
        got = code.decode_ms_batch(llrs, maxiters, variant=64)
        again = code.decode_ms_batch(llrs, maxiters, variant=64)
    torch.cuda.synchronize()
    m = 0 if all(bool((a == b).all()) and bool((a == c).all()) for a, b, c in zip(ref, got, again)) else 1
    k = min(n, 48)
    pick = rng.choice(n, k, replace=False)
    oc, ic, kc, _ = oracle.decode_ms_batch(code, pool[idx[pick]], maxiters)
    o = got[0][pick].cpu().numpy(); i = got[1][pick].cpu().numpy(); s = got[2][pick].cpu().numpy()
    mo = int(((o != oc).any(axis=1) | (i != ic) | (s != kc)).sum())
    total += n; checked += k; bad += m + mo
    if m or mo or t % 20 == 0:
        print(f"trial {t}: {code.name} {n} frames, max_iters {maxiters}: refill == lockstep {'no' if m else 'yes'}, oracle sample mismatches {mo}", flush=True)
print(f"TOTAL {trials} launches x 2, {total} frames against the lockstep kernel, {checked} against the oracle, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
