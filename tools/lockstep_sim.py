"""What the lockstep of the G codewords of a bit-sliced wave costs, and what a "straggler hand-off" could return, from MEASURED per-frame
pass counts (iters + success of variant 64 on device-generated AWGN frames: run with `collect` on the GPU first).
  python tools/lockstep_sim.py collect     # on the GPU box -> gpurun_out/passes/<code>_<Eb/N0>.npy
  python tools/lockstep_sim.py             # anywhere: the table of profiles/r05_kbench/lockstep.txt
Hand-off = at pass `cut` a group whose finished codewords number at least `need` stops; its unfinished frames are decoded again, from
scratch and bit-identically, in a second launch where stragglers share waves with stragglers."""
import sys, os
import numpy as np
CASES = (("TM5120", 4.0, 4), ("TM5120", 3.5, 4), ("TM2048", 2.5, 4), ("TM2048", 2.0, 4), ("TM1280", 4.0, 16), ("TM1536", 3.0, 8), ("TM6144", 3.0, 2))
OVH = 0.35                                      # prologue + epilogue of a group in iteration-equivalents (measured: 0.3-0.35)
if sys.argv[1:] == ["collect"]:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from labrador_ldpc_amd import LDPCCode
    os.makedirs("gpurun_out/passes", exist_ok=True)
    for name, ebn0, G in CASES:
        code = LDPCCode[name]
        rng = np.random.default_rng(1)
        pool = np.zeros((64, code.n() // 8), np.uint8)
        for i in range(64):
            code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
        sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
        llrs8 = code.awgn_frames(torch.from_numpy(pool).to(torch.device("cuda", 0)), 262144, sigma, seed=5, dtype="i8")
        out = code.decode_ms_batch(llrs8, 25, variant=64)
        np.save(f"gpurun_out/passes/{name}_{ebn0}.npy", (out[1].cpu().numpy().astype(np.int16) + out[2].cpu().numpy()).astype(np.int8))
    sys.exit(0)
def handoff(p, G, cut, need):
    g = p.reshape(-1, G)
    gmax = g.max(axis=1)
    hand = (gmax > cut) & ((g <= cut).sum(axis=1) >= need)
    first = (np.where(hand, cut, gmax) + OVH).sum()
    strag = g[hand][g[hand] > cut]
    n = len(strag) // G * G
    second = (strag[:n].reshape(-1, G).max(axis=1) + OVH).sum() if n else 0.0
    if len(strag) > n: second += strag[n:].max() + OVH
    return first + second, hand.mean(), len(strag) / p.size
print("code  Eb/N0   G   group passes / mean frame passes   best hand-off (cut, need)   gain   groups handing off   frames decoded twice")
for name, ebn0, G in CASES:
    p = np.load(f"gpurun_out/passes/{name}_{ebn0}.npy").astype(np.int64)
    now = (p.reshape(-1, G).max(axis=1) + OVH).sum()
    best = min(((*handoff(p, G, cut, need), cut, need) for cut in range(4, 25) for need in range(1, G)), key=lambda t: t[0]) if G > 1 else None
    print(f"{name} {ebn0:4.1f} {G:3d}   {p.reshape(-1, G).max(axis=1).mean() / p.mean():6.3f}" +
          (f"   ({best[3]}, {best[4]})   {now / best[0] - 1:+6.1%}   {best[1]:6.1%}   {best[2]:6.2%}" if best and best[1] > 0 else "   none pays"))
