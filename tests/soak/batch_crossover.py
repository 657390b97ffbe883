"""From how many frames per call does the GPU entry beat a loop of CPU calls?  (Round 3's review, weak #8: one TC128 frame takes 21.7 us
through labrador_ldpc_decode_ms_f32 and 9.9 us on the CPU.)  Per code, f32 LLRs at the code's working Eb/N0, 25 iterations max: the CPU
oracle's time per frame on ONE core (a stand-in for the reference's Rust decode_ms, which cannot be built here) against ONE
labrador_ldpc_decode_ms_batch_f32 call with host buffers (PCIe copies included) for B = 1, 2, 4 ... frames; the crossover is the smallest B
with t_gpu(B) < B * t_cpu.  Lives under tests/ because it runs the oracle.     python tests/soak/batch_crossover.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from labrador_ldpc_amd import LDPCCode

rng = np.random.default_rng(2)
print("code    Eb/N0  CPU us/frame   GPU us per call for B = 1, 2, 4, 8, 16, 32, 64, 256              GPU wins from")
for code in LDPCCode:
    ebn0 = 5.0 if code.n() <= 512 else (4.0 if code.k() * 5 == code.n() * 4 else 3.0 if code.k() * 3 == code.n() * 2 else 2.5)
    llrs, _ = oracle.awgn_llrs(code, rng, 256, ebn0, np.float32)
    oracle.decode_ms_batch(code, llrs[:8], 25, 1)
    t = time.perf_counter()
    oracle.decode_ms_batch(code, llrs[:64], 25, 1)
    cpu = (time.perf_counter() - t) / 64
    gpu, wins = [], None
    for B in (1, 2, 4, 8, 16, 32, 64, 256):
        l = llrs[:B]
        for _ in range(3):
            code.decode_ms_batch(l, 25)
        reps = 50
        t = time.perf_counter()
        for _ in range(reps):
            code.decode_ms_batch(l, 25)
        g = (time.perf_counter() - t) / reps
        gpu.append(g)
        if wins is None and g < B * cpu:
            wins = B
    print(f"{code.name:7s} {ebn0:4.1f} {cpu * 1e6:10.1f}     " + " ".join(f"{g * 1e6:7.1f}" for g in gpu) + f"      B >= {wins}")
