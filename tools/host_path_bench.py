"""PCIe-inclusive rate of the host-pointer entry points (LABRADOR_LDPC_HIP_MEM_HOST).

    python tools/host_path_bench.py [frames] [ebn0_db]

Generates noisy frames on the GPU, copies them to ordinary (pageable) numpy arrays, then times
labrador_ldpc_decode_ms_batch_{f32,i8} on those host arrays and checks the results against the
device-resident call on the same frames.  Not part of bench.py's `value` (inputs there are
resident in HBM when the timed region starts).
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from labrador_ldpc_amd import LDPCCode  # noqa: E402


def run(code, dtype, frames, ebn0_db, maxiters=25):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    rate = code.k() / code.n()
    sigma = float(np.sqrt(1.0 / (2.0 * rate * 10.0 ** (ebn0_db / 10.0))))
    llrs_d = code.awgn_frames(cws, frames, sigma, seed=99, dtype=dtype)
    out_d, it_d, ok_d = code.decode_ms_batch(llrs_d, maxiters=maxiters)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out_d, it_d, ok_d = code.decode_ms_batch(llrs_d, maxiters=maxiters)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t
    llrs_h = llrs_d.cpu().numpy()
    del llrs_d
    best = None
    for _ in range(3):
        t = time.perf_counter()
        out_h, it_h, ok_h = code.decode_ms_batch(llrs_h, maxiters=maxiters)
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    same = (np.array_equal(out_h, out_d.cpu().numpy()) and np.array_equal(it_h.astype(np.int64), it_d.cpu().numpy().astype(np.int64))
            and np.array_equal(ok_h, ok_d.cpu().numpy()))
    gb = llrs_h.nbytes / 1e9
    print(f"{code.name} {dtype}: {frames} frames ({gb:.2f} GB of LLRs)  host-pointer call {best*1e3:.1f} ms = "
          f"{frames/best/1e6:.3f} M frames/s ({gb/best:.1f} GB/s of LLRs); device-resident {frames/t_dev/1e6:.3f} M frames/s; "
          f"identical results: {same}", flush=True)


if __name__ == "__main__":
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    ebn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    run(LDPCCode.TM8192, "f32", frames, ebn0)
    run(LDPCCode.TM8192, "i8", frames, ebn0 + 1.0)
    run(LDPCCode.TM2048, "f32", frames * 2, ebn0 + 1.0)
    run(LDPCCode.TC512, "i8", frames * 8, 5.0)
