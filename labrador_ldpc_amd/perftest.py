"""BER Monte-Carlo of the min-sum decoder -- the GPU counterpart of the reference's perftest binary.

Reference: perftest/src/main.rs.  There every rayon worker loops `ms_trial` (:9-29): random bytes ->
encode -> hard_to_llrs (+-1) -> add Normal(0, sigma) noise -> decode_ms (100 iterations) -> count bit
errors in the first k bits, until trials*k > 5e7 or errors > 5000 (:50); one CSV line per SNR (:62):
    code,snr,trials,bits,errors,ber
Here a trial batch runs entirely on the device: labrador_ldpc_encode_batch -> labrador_ldpc_hip_awgn_f32
-> labrador_ldpc_decode_ms_batch_f32, errors counted with torch bit ops (plumbing).

Noise conventions:
  --noise perftest  sigma = 10^(-snr_db/10), what the reference calls "snr" (perftest/src/main.rs:15)
  --noise ebn0      sigma^2 = 1 / (2 R 10^(EbN0/10)), R = k/n (textbook Eb/N0; what bench.py uses)

    python -m labrador_ldpc_amd.perftest --code TC512 --noise perftest
"""
from __future__ import annotations

import argparse
import sys

import numpy as np


def sigma_for(code, snr_db: float, noise: str) -> float:
    if noise == "perftest":
        return float(1.0 / 10.0 ** (snr_db / 10.0))
    return float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (snr_db / 10.0))))


def ms_trials(code, snr_db: float, noise: str = "perftest", maxiters: int = 100, batch: int = 65536,
              max_bits: float = 5e7, max_errors: int = 5000, seed: int = 1, device: int = 0):
    """One SNR point.  Returns (trials, bits, errors, ber, frame_errors)."""
    import torch
    dev = torch.device("cuda", device)
    k8 = code.k() // 8
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = sigma_for(code, snr_db, noise)
    popcnt = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    trials = errors = frame_errors = 0
    rounds = 0
    while trials * code.k() <= max_bits and errors <= max_errors:
        data = torch.randint(0, 256, (batch, k8), dtype=torch.uint8, device=dev, generator=g)
        cw = code.encode_batch(data)                                         # perftest/src/main.rs:10-12
        llrs = code.awgn_frames(cw, batch, sigma, seed=(seed << 20) + rounds)  # :13-18 (frame f <- codeword f)
        out, _, _ = code.decode_ms_batch(llrs, maxiters)                    # :22
        diff = out[:, :k8] ^ data                                            # :23-28
        per_frame = popcnt[diff.long()].sum(dim=1)
        errors += int(per_frame.sum())
        frame_errors += int((per_frame > 0).sum())
        trials += batch
        rounds += 1
    bits = trials * code.k()
    ber = max(1, errors) / bits                                              # :59-61
    return trials, bits, errors, ber, frame_errors


def main(argv=None):
    from . import LDPCCode
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--code", default="TC512")                               # perftest/src/main.rs:69
    ap.add_argument("--snrs", default="0.8,0.9,1.0,1.1,1.2,1.3,1.4,1.5,1.6,1.7,1.8,1.9,2.0,2.1,2.2")   # :68
    ap.add_argument("--noise", choices=["perftest", "ebn0"], default="perftest")
    ap.add_argument("--maxiters", type=int, default=100)                     # :22
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--max-bits", type=float, default=5e7)
    ap.add_argument("--max-errors", type=int, default=5000)
    args = ap.parse_args(argv)
    code = LDPCCode[args.code]
    for snr in (float(x) for x in args.snrs.split(",")):
        trials, bits, errors, ber, fe = ms_trials(code, snr, args.noise, args.maxiters, args.batch,
                                                  args.max_bits, args.max_errors)
        print(f"{code.name},{snr:.2f},{trials},{bits},{max(1, errors)},{ber:.5e}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
