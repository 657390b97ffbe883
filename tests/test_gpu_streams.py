"""Boundary behaviour of the batched entry points: non-default HIP streams, device selection through
opts, host batches larger than one staging chunk, misaligned device buffers."""
import ctypes

import numpy as np
import pytest

import oracle
import labrador_ldpc_amd as la
from labrador_ldpc_amd import LDPCCode

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_non_default_stream_is_honoured():
    code = LDPCCode.TM2048
    rng = np.random.default_rng(1)
    llrs, _ = oracle.awgn_llrs(code, rng, 256, 2.5, np.float32)
    ref_out, ref_it, ref_ok, _ = oracle.decode_ms_batch(code, llrs, 25)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        d = torch.from_numpy(llrs).to(dev, non_blocking=True)
        out, it, ok = code.decode_ms_batch(d, 25)           # picks torch's current stream = s
        ev = torch.cuda.Event()
        ev.record(s)
    ev.synchronize()
    assert (out.cpu().numpy() == ref_out).all() and (it.cpu().numpy() == ref_it).all() and (ok.cpu().numpy() == ref_ok).all()
    # explicit stream handle
    out2, it2, ok2 = code.decode_ms_batch(d, 25, stream=s.cuda_stream)
    s.synchronize()
    assert torch.equal(out2, out) and torch.equal(it2, it)


def test_host_batch_larger_than_one_staging_chunk():
    """MEM_HOST calls stage 65 536 frames at a time (csrc/capi.hip)."""
    code = LDPCCode.TC128
    rng = np.random.default_rng(2)
    base, _ = oracle.awgn_llrs(code, rng, 1024, 3.0, np.float32)
    llrs = np.tile(base, (70, 1))[: 65536 + 4097]
    out, it, ok = code.decode_ms_batch(llrs, 25)
    ref_out, ref_it, ref_ok, _ = oracle.decode_ms_batch(code, base, 25)
    idx = np.arange(len(llrs)) % 1024
    assert (out == ref_out[idx]).all() and (it == ref_it[idx]).all() and (ok == ref_ok[idx]).all()


def test_misaligned_device_output_is_rejected():
    code = LDPCCode.TC128
    dev = torch.device("cuda", 0)
    llrs = torch.zeros((4, code.n()), dtype=torch.float32, device=dev)
    buf = torch.empty(4 * code.output_len() + 8, dtype=torch.uint8, device=dev)
    it = torch.empty(4, dtype=torch.int32, device=dev)
    ok = torch.empty(4, dtype=torch.uint8, device=dev)
    opts = la.HipOpts(0, la.MEM_DEVICE, None, 0)
    st = la.lib.labrador_ldpc_decode_ms_batch_f32(int(code), llrs.data_ptr(), buf.data_ptr() + 1, it.data_ptr(),
                                                  ok.data_ptr(), 4, 10, ctypes.byref(opts))
    assert st == -1 and "aligned" in la.last_error()


def test_bad_device_and_variant_are_reported():
    code = LDPCCode.TC128
    llrs = np.zeros((2, code.n()), dtype=np.float32)
    out = np.zeros((2, code.output_len()), dtype=np.uint8)
    it = np.zeros(2, dtype=np.uint32)
    ok = np.zeros(2, dtype=np.uint8)
    opts = la.HipOpts(99, la.MEM_HOST, None, 0)
    st = la.lib.labrador_ldpc_decode_ms_batch_f32(int(code), llrs.ctypes.data, out.ctypes.data, it.ctypes.data,
                                                  ok.ctypes.data, 2, 10, ctypes.byref(opts))
    assert st == -1 and "out of range" in la.last_error()
    opts = la.HipOpts(-1, la.MEM_HOST, None, 7)
    st = la.lib.labrador_ldpc_decode_ms_batch_f32(int(code), llrs.ctypes.data, out.ctypes.data, it.ctypes.data,
                                                  ok.ctypes.data, 2, 10, ctypes.byref(opts))
    assert st == -4 and "variant" in la.last_error()
