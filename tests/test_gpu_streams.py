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
    """MEM_HOST calls stage at most 262 144 frames at a time (csrc/capi.hip chunk_items)."""
    code = LDPCCode.TC128
    rng = np.random.default_rng(2)
    base, _ = oracle.awgn_llrs(code, rng, 1024, 3.0, np.float32)
    llrs = np.tile(base, (261, 1))[: 262144 + 4097]
    out, it, ok = code.decode_ms_batch(llrs, 25)
    ref_out, ref_it, ref_ok, _ = oracle.decode_ms_batch(code, base, 25)
    idx = np.arange(len(llrs)) % 1024
    assert (out == ref_out[idx]).all() and (it == ref_it[idx]).all() and (ok == ref_ok[idx]).all()


@pytest.fixture
def small_chunks(monkeypatch):
    monkeypatch.setenv("LABRADOR_LDPC_HIP_CHUNK", "300")      # read by the library on every host-pointer call
    yield 300


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16, np.float64])
@pytest.mark.parametrize("frames", [300, 301, 599, 600, 1501])
def test_host_pipeline_decode_ms_matches_oracle(small_chunks, dtype, frames):
    """1, 2 (one ragged), 2, and 6 chunks through the three-stream host pipeline; every frame distinct."""
    code = LDPCCode.TM1280
    rng = np.random.default_rng(frames)
    llrs, _ = oracle.awgn_llrs(code, rng, frames, 3.5, dtype)
    out, it, ok = code.decode_ms_batch(llrs, 20)
    ref_out, ref_it, ref_ok, _ = oracle.decode_ms_batch(code, llrs, 20)
    assert (out == ref_out).all() and (it == ref_it).all() and (ok == ref_ok).all()


def test_host_pipeline_on_a_caller_stream_and_from_two_threads(small_chunks):
    import threading
    code = LDPCCode.TM1536
    rng = np.random.default_rng(77)
    llrs, _ = oracle.awgn_llrs(code, rng, 1000, 3.0, np.float32)
    ref = oracle.decode_ms_batch(code, llrs, 20)
    s = torch.cuda.Stream(device=torch.device("cuda", 0))
    out, it, ok = code.decode_ms_batch(llrs, 20, stream=s.cuda_stream)
    assert (out == ref[0]).all() and (it == ref[1]).all() and (ok == ref[2]).all()
    results = [None, None]

    def work(k):
        results[k] = code.decode_ms_batch(llrs[k::2].copy(), 20)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(2):
        assert (results[k][0] == ref[0][k::2]).all() and (results[k][1] == ref[1][k::2]).all()


def test_host_pipeline_decode_bf_and_encode(small_chunks):
    code = LDPCCode.TC256
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=(1000, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(data)
    assert cws.shape == (1000, code.n() // 8)
    for f in (0, 299, 300, 999):
        assert (cws[f] == oracle.copy_encode(code, data[f])).all()
    rx = cws.copy()
    flips = rng.integers(0, code.n(), size=1000)
    rx[np.arange(1000), flips // 8] ^= (0x80 >> (flips % 8)).astype(np.uint8)
    out, it, ok = code.decode_bf_batch(rx, 30)
    for f in (0, 1, 299, 300, 301, 998, 999):
        r_ok, r_it, r_out = oracle.decode_bf(code, rx[f], 30)
        assert ok[f] == r_ok and it[f] == r_it and (out[f] == r_out).all()
    good = ok == 1
    assert good.mean() > 0.9 and (out[good][:, : code.k() // 8] == data[good]).all()


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


def test_device_calls_can_be_captured_in_a_hip_graph():
    """Device-resident calls only enqueue work on the given stream, so a channel + decode pass can be
    captured once into a HIP graph and replayed (the launch-bound small-batch case)."""
    code = LDPCCode.TM1280
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, size=(64, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(torch.from_numpy(data).to(dev))
    frames = 512
    llrs = torch.empty((frames, code.n()), dtype=torch.float32, device=dev)
    out = torch.empty((frames, code.output_len()), dtype=torch.uint8, device=dev)
    it = torch.empty((frames,), dtype=torch.int32, device=dev)
    ok = torch.empty((frames,), dtype=torch.uint8, device=dev)
    sigma = 0.55

    def one_pass():
        code.awgn_frames(cws, frames, sigma, seed=1234, dtype="f32", out=llrs)
        code.decode_ms_batch(llrs, 25, output=out, iters=it, success=ok)

    one_pass()                                   # warm-up outside the capture (module load, occupancy query)
    torch.cuda.synchronize()
    ref_llrs = llrs.cpu().numpy().copy()
    ref = oracle.decode_ms_batch(code, ref_llrs, 25)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one_pass()
    for _ in range(3):
        llrs.zero_(); out.zero_(); it.zero_(); ok.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (llrs.cpu().numpy() == ref_llrs).all()
        assert (out.cpu().numpy() == ref[0]).all() and (it.cpu().numpy() == ref[1]).all() and (ok.cpu().numpy() == ref[2]).all()


def test_two_pass_nan_decode_in_a_hip_graph_and_in_slices(monkeypatch):
    """TM5120 f32 above 1 024 frames is TWO launches per call (decode_ms_launch.hpp, two_pass_nan(): a NaN-blind kernel that
    marks, then the NaN-handling kernel over the marked codewords).  Both are plain stream work: captured into a HIP graph and
    replayed they reproduce the oracle, NaN frames included, and so do the slices of a batch larger than one launch."""
    code = LDPCCode.TM5120
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(23)
    base, _ = oracle.awgn_llrs(code, rng, 48, 3.6, np.float32)
    base.view(np.uint32)[40:, 7] = 0xFFC00000                  # eight frames with a (negative, quiet) NaN LLR
    base.view(np.uint32)[44:, ::3] = 0x7FA00000                # four of them mostly signalling NaNs
    ref = oracle.decode_ms_batch(code, base, 25)
    frames = 1536
    idx = rng.integers(0, len(base), frames)
    idx[0], idx[-1] = 47, 40
    llrs = torch.from_numpy(base[idx]).to(dev)
    out = torch.empty((frames, code.output_len()), dtype=torch.uint8, device=dev)
    it = torch.empty((frames,), dtype=torch.int32, device=dev)
    ok = torch.empty((frames,), dtype=torch.uint8, device=dev)

    def check():
        torch.cuda.synchronize()
        assert (out.cpu().numpy() == ref[0][idx]).all() and (it.cpu().numpy() == ref[1][idx]).all() and (ok.cpu().numpy() == ref[2][idx]).all()

    code.decode_ms_batch(llrs, 25, output=out, iters=it, success=ok)
    check()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        code.decode_ms_batch(llrs, 25, output=out, iters=it, success=ok)
    for _ in range(3):
        out.zero_(); it.fill_(-1); ok.zero_()                  # (-1 = the mark itself: a stale one must not survive)
        g.replay()
        check()
    monkeypatch.setenv("LABRADOR_LDPC_HIP_MAX_LAUNCH", "1100")  # slices of 1100 and 436 frames: two passes, then one
    out.zero_(); it.fill_(-1); ok.zero_()
    code.decode_ms_batch(llrs, 25, output=out, iters=it, success=ok)
    check()


def test_device_batches_larger_than_one_launch_are_sliced(monkeypatch):
    """The kernels take 32-bit frame counts; a device-resident batch is enqueued as launches of at most 2^30
    frames (no size_t batch may be truncated).  The slice is lowered here so that a small batch
    exercises the slicing arithmetic (pointer offsets of all four buffers, ragged last slice, TC128 packs four
    codewords per wave) for decode_ms, decode_bf and the encoder."""
    monkeypatch.setenv("LABRADOR_LDPC_HIP_MAX_LAUNCH", "1000")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    for code, frames in ((LDPCCode.TC128, 3503), (LDPCCode.TM1280, 2077)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, 4.0, np.float32)
        out, it, ok = code.decode_ms_batch(torch.from_numpy(llrs).to(dev), 20)
        ref = oracle.decode_ms_batch(code, llrs, 20)
        assert (out.cpu().numpy() == ref[0]).all() and (it.cpu().numpy() == ref[1]).all() and (ok.cpu().numpy() == ref[2]).all()
        data = rng.integers(0, 256, size=(frames, code.k() // 8), dtype=np.uint8)
        cws = code.encode_batch(torch.from_numpy(data).to(dev))
        monkeypatch.delenv("LABRADOR_LDPC_HIP_MAX_LAUNCH")
        assert torch.equal(cws, code.encode_batch(torch.from_numpy(data).to(dev)))
        monkeypatch.setenv("LABRADOR_LDPC_HIP_MAX_LAUNCH", "1000")
        rx = cws.clone()
        rx[:, 1] ^= 0x08
        o_b, i_b, k_b = code.decode_bf_batch(rx, 30)
        assert bool((k_b == 1).all()) and torch.equal(o_b[:, : code.n() // 8], cws)


def test_single_frame_calls_with_and_without_the_completion_ticket():
    """The reference-shaped single-frame entries on the small codes do not synchronise their stream: the kernel (its notifying twin,
    csrc/notify.hpp) stores the call's ticket into pinned memory and the calling thread spins on it.  Hundreds of calls in a row, of
    codes and types whose calls are notified (TC128 ... TM2048 with once-read LLRs) interleaved with calls that are not (larger codes;
    the register-lean f32 kernels, whose input is copied; two frames per call), must return what the oracle does -- the ticket of call
    k must never be taken for call k + 1's -- and the same with LABRADOR_LDPC_HIP_NO_NOTIFY=1 in a process of its own."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(91)
    jobs = []
    for code, dtype, ebn0 in ((LDPCCode.TC128, np.float32, 3.0), (LDPCCode.TC512, np.int8, 3.0), (LDPCCode.TM2048, np.float32, 2.0),
                              (LDPCCode.TM1280, np.float32, 3.5), (LDPCCode.TM5120, np.int8, 3.5), (LDPCCode.TC256, np.float64, 3.0),
                              (LDPCCode.TM1536, np.int16, 2.5), (LDPCCode.TM8192, np.float32, 2.0)):
        llrs, _ = oracle.awgn_llrs(code, rng, 6, ebn0, dtype)
        jobs.append((code, llrs, oracle.decode_ms_batch(code, llrs, 30)))
    for rep in range(40):
        for code, llrs, (oc, ic, kc, _) in jobs:
            f = (rep * 5 + int(code)) % 6
            out = np.full(code.output_len(), 0xEE, np.uint8)
            ok, iters = code.decode_ms(llrs[f], out, maxiters=30)
            assert bool(ok) == bool(kc[f]) and (out == oc[f]).all() and (not ok or iters == ic[f]), (code.name, rep, f)
        if rep % 7 == 3:                                     # a two-frame host batch in between: one launch, not notified
            code, llrs, (oc, ic, kc, _) = jobs[rep % len(jobs)]
            o, i, k = code.decode_ms_batch(llrs[:2], 30)
            assert (o == oc[:2]).all() and (i == ic[:2]).all() and (k == kc[:2]).all()
    if os.environ.get("LABRADOR_LDPC_HIP_NO_NOTIFY"):
        return
    env = dict(os.environ, LABRADOR_LDPC_HIP_NO_NOTIFY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__ + "::test_single_frame_calls_with_and_without_the_completion_ticket"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
