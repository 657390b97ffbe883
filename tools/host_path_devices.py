"""PCIe-inclusive rate of the host-pointer entry points through the device-set path (worker thread per listed device, pinned to
the CPUs local to its GPU), against the direct single-device call and with the pinning switched off.
    python tools/host_path_devices.py            # runs itself twice: pinned workers, LABRADOR_LDPC_HIP_NO_NUMA=1"""
import os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from labrador_ldpc_amd import LDPCCode
    code, frames = LDPCCode.TM8192, 131072
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=(256, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    llrs = code.awgn_frames(cws, frames, 0.7943, seed=99, dtype="f32").cpu().numpy()
    out = np.zeros((frames, code.output_len()), np.uint8); it = np.zeros(frames, np.uint32); ok = np.zeros(frames, np.uint8)
    ref = None
    for name, devices in (("direct call, device 0", None), ("device list [0]: one pinned worker", [0]),
                          ("device list [0, 0]: two pipelines on the GPU", [0, 0])):
        best = 1e9
        for _ in range(4):
            t = time.perf_counter()
            code.decode_ms_batch(llrs, 25, output=out, iters=it, success=ok, devices=devices)
            best = min(best, time.perf_counter() - t)
        if ref is None:
            ref = (out.copy(), it.copy(), ok.copy())
        same = (out == ref[0]).all() and (it == ref[1]).all() and (ok == ref[2]).all()
        print(f"  {name:48s} {frames / best / 1e6:6.3f} M frames/s  {llrs.nbytes / best / 1e9:5.1f} GB/s of LLRs  results {'identical' if same else 'DIFFER'}", flush=True)
    with open("/proc/self/status") as f:
        print("  main thread", [l.strip() for l in f if l.startswith("Cpus_allowed_list")][0])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for label, env in (("workers pinned to the GPU's local CPUs", {}), ("LABRADOR_LDPC_HIP_NO_NUMA=1", {"LABRADOR_LDPC_HIP_NO_NUMA": "1"})):
            print(f"TM8192 f32, 131072 frames from ordinary (pageable) numpy arrays, 2 dB, 25 iterations -- {label}:", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=False)
        for p in sorted(os.listdir("/sys/bus/pci/devices")) if os.path.isdir("/sys/bus/pci/devices") else []:
            d = os.path.join("/sys/bus/pci/devices", p)
            try:
                if open(os.path.join(d, "class")).read().startswith("0x0302") or open(os.path.join(d, "class")).read().startswith("0x0380"):
                    print(f"{p}: numa_node {open(os.path.join(d, 'numa_node')).read().strip()}, local_cpulist {open(os.path.join(d, 'local_cpulist')).read().strip()}")
            except OSError:
                pass
