"""One kernel or two for NaN LLRs (decode_ms_launch.hpp, two_pass_nan()): library-level rates of TM5120 f32 with
`variant` 512 (one pass: the NaN-handling kernel), 1024 (two passes: the NaN-blind kernel that marks, then the NaN-handling kernel
over the marked codewords) and 0 (the launcher's choice), over batch sizes, clean and with a share of NaN frames.
    python tools/nan_two_pass_ab.py > gpurun_out/nan_two_pass_ab.txt"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)


def rate(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = max(3, int(0.05 / max(1e-6, n / 60e6)))
    reps = min(reps, 200)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    return n / best / 1e6


print("code    frames  nan_share   one_pass   two_pass    default   M codewords/s")
for code, eb in ((LDPCCode.TM5120, 4.0), (LDPCCode.TM1280, 4.0)):
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (eb / 10.0))))
    for frames in (256, 512, 1024, 2048, 4096, 16384, 65536, 262144, 524288) if code == LDPCCode.TM5120 else (4096, 16384, 65536, 262144, 1048576):
        for share in (0.0, 0.01) if frames >= 65536 else (0.0,):
            llrs = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32")
            if share:
                rows = torch.from_numpy(rng.choice(frames, int(frames * share), replace=False)).to(dev)
                llrs[rows, 17] = float("nan")
            res = {}
            outs = {}
            for name, variant in (("one", 512), ("two", 1024), ("def", 0)):
                outs[name] = code.decode_ms_batch(llrs, 25, variant=variant)
                res[name] = rate(lambda: code.decode_ms_batch(llrs, 25, variant=variant), frames)
            same = all(torch.equal(a, b) for a, b in zip(outs["one"], outs["two"])) and all(torch.equal(a, b) for a, b in zip(outs["one"], outs["def"]))
            print(f"{code.name:7s} {frames:7d} {share:9.2f} {res['one']:10.2f} {res['two']:10.2f} {res['def']:10.2f}   {'identical' if same else 'RESULTS DIFFER'}", flush=True)
            del llrs
