"""Round 3, item 1: BASELINE config 4 whole (4 194 304 TM8192 f32 frames, 137 GB) on ONE GPU against its eight 524 288-frame
slices, same process, same buffers: whole launch vs slice launches, queue-fed vs fixed-stride distribution.
    python tools/r03_fullbatch.py [total_frames] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from labrador_ldpc_amd import LDPCCode
from labrador_ldpc_amd.sharding import frame_seed

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
code = LDPCCode.TM8192
dev = torch.device("cuda", 0)
sigma = float(np.sqrt(1.0 / (2.0 * 0.5 * 10.0 ** 0.2)))
rng = np.random.default_rng(0x1DBC + int(code))
cws = np.zeros((256, code.n() // 8), dtype=np.uint8)
for i in range(256):
    code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), cws[i])
t0 = time.perf_counter()
llrs = code.awgn_frames(torch.from_numpy(cws).to(dev), total, sigma, frame_seed(0x1DBC + int(code), 0), dtype="f32")
torch.cuda.synchronize()
print(f"generated {total} frames ({llrs.numel() * 4 / 1e9:.1f} GB) in {time.perf_counter() - t0:.1f} s", flush=True)
out = torch.empty((total, code.output_len()), dtype=torch.uint8, device=dev)
it = torch.empty((total,), dtype=torch.int32, device=dev)
ok = torch.empty((total,), dtype=torch.uint8, device=dev)


def timed(lo, hi, variant):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    code.decode_ms_batch(llrs[lo:hi], 25, output=out[lo:hi], iters=it[lo:hi], success=ok[lo:hi], variant=variant)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


res = {"total_frames": total}
S = total // 8
timed(0, S, 0)                                               # warm-up
for name, variant in (("queue", 0), ("stride", 256)):
    whole = [timed(0, total, variant) for _ in range(reps)]
    slices = [[timed(s * S, (s + 1) * S, variant) for s in range(8)] for _ in range(reps)]
    res[name] = {"whole_ms": whole, "whole_Mcw_s": [total / m / 1e3 for m in whole],
                 "slice_ms": slices, "slices_sum_ms": [sum(r) for r in slices],
                 "slice_Mcw_s": [[S / m / 1e3 for m in r] for r in slices]}
    print(name, json.dumps(res[name]), flush=True)
res["mean_iters"] = float(it.double().mean())
res["fail_rate"] = 1.0 - float(ok.double().mean())
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r03_fullbatch.json"), "w") as f:
    json.dump(res, f, indent=1)
