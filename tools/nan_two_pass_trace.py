"""Workload for a rocprofv3 --kernel-trace of the two-launch TM5120 f32 decode: 524 288 frames, three decodes without a NaN, three
with a NaN in every hundredth frame (the kernels' template arguments end in 1 = first pass, 2 = second pass).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nan_trace -- python3 $R/tools/nan_two_pass_trace.py"""
import os, sys, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R)
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
code, frames = LDPCCode.TM5120, 524288
data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
cws = code.encode_batch(torch.from_numpy(data).to(dev))
sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** 0.4)))
llrs = code.awgn_frames(cws, frames, sigma, seed=5, dtype="f32")
for _ in range(4):
    code.decode_ms_batch(llrs, 25)
torch.cuda.synchronize()
rows = torch.from_numpy(rng.choice(frames, frames // 100, replace=False)).to(dev)
llrs[rows, 17] = float("nan")
for _ in range(4):
    code.decode_ms_batch(llrs, 25)
torch.cuda.synchronize()
