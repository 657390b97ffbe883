"""The kernels either side of decode_ms (encoder, hard-decision decoder, synthetic channel) on device-resident batches, for
rocprofv3 (tools/prof_frow.sh): a few launches each of encode_batch, decode_bf_batch (bit-sliced and, with LABRADOR_LDPC_HIP_BF_BYTES=1,
byte-per-variable) and awgn_frames for TM8192 and TM2048.  Prints the algorithmic bytes per launch of each so that the profile's durations
turn into fractions of the HBM roofline."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from labrador_ldpc_amd import LDPCCode
dev = torch.device("cuda", 0)
info = {}
for code in (LDPCCode.TM8192, LDPCCode.TM2048):
    frames = 262144 * 8192 // code.n()
    rng = np.random.default_rng(1)
    data = torch.from_numpy(rng.integers(0, 256, (4096, code.k() // 8), dtype=np.uint8)).to(dev)[torch.arange(frames, device=dev) % 4096].contiguous()
    for _ in range(3):
        cws = code.encode_batch(data)
    flips = torch.zeros_like(cws)
    flips[:, ::97] = 0x10                                      # a few bit errors per frame
    hard = cws ^ flips
    for _ in range(3):
        out = code.decode_bf_batch(hard, 50)
    sigma = float(np.sqrt(1.0 / (2.0 * 0.5 * 10.0 ** 0.2)))
    for _ in range(3):
        llrs = code.awgn_frames(cws[:256].contiguous(), frames, sigma, seed=3, dtype="f32")
    for _ in range(3):
        l8 = code.awgn_frames(cws[:256].contiguous(), frames, sigma, seed=3, dtype="i8")
    torch.cuda.synchronize()
    info[code.name] = {"frames": frames, "encode_bytes": frames * (code.k() // 8 + code.n() // 8),
                       "decode_bf_bytes": frames * (code.n() // 8 + code.output_len() + 5), "awgn_f32_bytes": frames * code.n() * 4,
                       "awgn_i8_bytes": frames * code.n(), "bf_mean_iters": float(out[1].double().mean()), "bf_success": float(out[2].double().mean())}
print(json.dumps(info))
