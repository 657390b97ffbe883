"""The bit-sliced i8 kernel (csrc/decode_ms_bitslice.hpp, `variant` 64) on the GPU against the oracle and the frozen i8 golden
files: the CPU tests (tests/test_bitslice_emu.py) validate the formulation by running the kernel's source text lane by lane;
here gfx950 executes the same text -- v_bitop3_b32, ds_bpermute_b32, v_alignbit_b32 -- and must return the same bytes, through the
device-pointer and the host-pointer entry points, for whole and part-filled waves, and equal to the default i8 kernels."""
import glob
import os

import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 64                                                       # `variant` of the bit-sliced kernel
CODES = [LDPCCode.TM1280, LDPCCode.TM1536, LDPCCode.TM2048, LDPCCode.TM5120, LDPCCode.TM6144, LDPCCode.TM8192]
# (the rate-4/5 codes' bit-sliced kernel shares a codeword group between two waves: csrc/decode_ms_bitslice_split.hpp)
# `variant` 64 on TM1536 and TM1280 is the SLOT-REFILL kernel (decode_refill / decode_refill_split: a finished slot takes the wave's next
# frame at once; the default dispatch picks it from 65 536 frames up); 64 | 256 names their lockstep kernels, which smaller default
# batches and streams without a queue word run: both must return the oracle's bytes
KERNELS = [(c, BS) for c in CODES] + [(LDPCCode.TM1536, BS | 256), (LDPCCode.TM1280, BS | 256)]
KERNEL_IDS = [f"{c.name}-{v}" for c, v in KERNELS]


def _same(code, llrs, maxiters, variant=BS):
    o, i, k = code.decode_ms_batch(llrs, maxiters, variant=variant)
    oc, ic, kc, _ = oracle.decode_ms_batch(code, llrs, maxiters)
    bad = np.nonzero((o != oc).any(axis=1) | (i != ic) | (k != kc))[0]
    assert bad.size == 0, f"frames {bad.tolist()[:8]} differ (iters {i[bad][:8].tolist()} vs {ic[bad][:8].tolist()})"
    return i, k


@pytest.mark.parametrize("code,variant", KERNELS, ids=KERNEL_IDS)
def test_bitsliced_kernel_equals_the_oracle(code, variant):
    rng = np.random.default_rng(500 + int(code))
    hi = 1.5 if code.k() * 5 == code.n() * 4 else 0.0                 # the rate-4/5 codes converge 1.5 dB later
    for ebn0, scale, lim, frames in ((2.0 + hi, 8.0, 31, 257), (1.0, 30.0, 127, 33), (4.5 + hi, 16.0, 127, 64), (2.5 + hi, 8.0, 31, 1)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, np.int8, scale=scale, lim=lim)
        for maxiters in (25, 3, 0, 60):
            _same(code, llrs, maxiters, variant)


@pytest.mark.parametrize("code,variant", KERNELS, ids=KERNEL_IDS)
def test_bitsliced_kernel_on_corner_inputs(code, variant):
    N = code.n()
    rng = np.random.default_rng(11)
    frames = [np.zeros(N, np.int8), np.full(N, -128, np.int8), np.full(N, 127, np.int8), rng.integers(-128, 128, N).astype(np.int8),
              rng.choice(np.array([-128, 127, 0, 1, -1], np.int8), N)]
    base, _ = oracle.awgn_llrs(code, rng, 3, 3.0, np.int8, scale=60.0, lim=127)
    spiked = base.copy()
    spiked[:, ::7] = -128
    llrs = np.stack(frames + list(spiked) + list(base))
    for maxiters in (25, 1, 2):
        _same(code, llrs, maxiters, variant)


def test_bitsliced_kernel_reproduces_the_i8_golden_files():
    for code, variant in KERNELS:
        z = np.load(os.path.join(ROOT, "tests", "golden", f"awgn_{code.name}_i8.npz"))
        for maxiters in (25, 4, 0):
            o, i, k = code.decode_ms_batch(z["llrs"], maxiters, variant=variant)
            assert (o == z[f"output_{maxiters}"]).all() and (i == z[f"iters_{maxiters}"]).all() and (k == z[f"success_{maxiters}"]).all()


@pytest.mark.parametrize("code", CODES)
def test_bitsliced_kernel_equals_the_default_i8_kernel_on_a_large_batch(code):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    pool = np.zeros((16, code.n() // 8), np.uint8)
    for i in range(16):
        code.copy_encode(rng.integers(0, 256, code.k() // 8, dtype=np.uint8), pool[i])
    ebn0 = 3.5 if code.k() * 5 == code.n() * 4 else (2.5 if code.k() * 3 == code.n() * 2 else 2.0)
    sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
    frames = 100003 if code.n() >= 5120 else 400003            # odd: the last wave is part-filled
    llrs = code.awgn_frames(torch.from_numpy(pool).to(dev), frames, sigma, seed=99, dtype="i8")
    old = 32 if code == LDPCCode.TM8192 else 1                  # the f32-pipe kernels by their explicit variant
    a = code.decode_ms_batch(llrs, 25, variant=old)
    b = code.decode_ms_batch(llrs, 25, variant=BS)
    c = code.decode_ms_batch(llrs, 25)                          # the default: bit-sliced from 768 ... 2048 groups up, by code
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and all(torch.equal(x, y) for x, y in zip(a, c))
    assert 0.5 < float(a[2].float().mean()) <= 1.0
    small = llrs[:777]                                          # below the threshold the default is the f32-pipe kernel: same results
    assert all(torch.equal(x, y[:777]) for x, y in zip(code.decode_ms_batch(small, 25), a))


def test_default_form_can_be_captured_into_a_graph():
    """No kernel of the default dispatch allocates anything (round 4's first rate-4/5 kernel drew a stream-ordered workspace and had to
    stay out of graph captures): a captured and replayed decode of a rate-4/5 batch above the bit-sliced threshold equals the oracle."""
    dev = torch.device("cuda", 0)
    code = LDPCCode.TM5120
    rng = np.random.default_rng(8)
    base, _ = oracle.awgn_llrs(code, rng, 40, 4.0, np.int8, scale=8.0, lim=31)
    ref = oracle.decode_ms_batch(code, base, 25)
    frames = 10243                                              # above the default's threshold (2048 groups of four), ragged last group
    idx = rng.integers(0, len(base), frames)
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
    for _ in range(2):
        out.zero_(); it.fill_(-1); ok.zero_()
        g.replay()
        check()


def test_variant_64_is_refused_where_it_does_not_exist():
    """The bit-sliced kernel is built for i8 on the TM codes: `variant` 64 on a TC code, or with an LLR buffer that is not 4-byte
    aligned, reports EUNSUPPORTED (never a silent other kernel); the default dispatch takes the f32-pipe kernel for such a buffer."""
    from labrador_ldpc_amd import LdpcHipError
    code = LDPCCode.TC512
    llrs = np.ones((64, code.n()), np.int8)
    with pytest.raises(LdpcHipError):
        code.decode_ms_batch(llrs, 5, variant=BS)
    code = LDPCCode.TM2048
    dev = torch.device("cuda", 0)
    raw = torch.ones(9000 * code.n() + 1, dtype=torch.int8, device=dev)
    odd = raw[1:].view(9000, code.n())                          # device buffer at an odd address
    with pytest.raises(LdpcHipError):
        code.decode_ms_batch(odd, 5, variant=BS)                # refused before anything is launched
    # ... and an OUTPUT buffer that is not 4-byte aligned (the kernel stores dwords; round 4 advice): the C entry wants 8-byte
    # alignment of device outputs anyway and says so
    even = torch.ones((9000, code.n()), dtype=torch.int8, device=dev)
    out_raw = torch.empty(9000 * code.output_len() + 2, dtype=torch.uint8, device=dev)
    out_odd = out_raw[2:].view(9000, code.output_len())
    with pytest.raises(LdpcHipError):
        code.decode_ms_batch(even, 5, output=out_odd, variant=BS)


@pytest.mark.parametrize("code", [LDPCCode.TM1536, LDPCCode.TM1280], ids=lambda c: c.name)
def test_slot_refill_hands_out_every_frame_exactly_once(code):
    """TM1536 (one wave per group) and TM1280 (two waves per group, both drawing the same chunks) through their slot-refill kernels on
    batches around their chunk and grid sizes -- 1 frame, fewer frames than a wave has slots, one frame more than a chunk, a batch that
    ends inside a chunk -- with frames that finish after 3 ... 25 iterations and frames that never do: every frame's results equal the
    lockstep kernel's (whose equality with the oracle the tests above pin), twice in a row on one stream (the queue word must be back
    at zero) and on a second stream."""
    rng = np.random.default_rng(77)
    hi = 1.5 if code is LDPCCode.TM1280 else 0.0
    pool = [oracle.awgn_llrs(code, rng, 512, e, np.int8, scale=s, lim=l)[0] for e, s, l in ((3.0 + hi, 8.0, 31), (1.0, 30.0, 127), (5.0 + hi, 8.0, 31))]
    llrs = np.concatenate(pool)
    rng.shuffle(llrs)
    for frames in (1, 7, 8, 9, 15, 16, 17, 63, 65, 1025, 1536):
        d = torch.from_numpy(llrs[:frames]).cuda()
        ref = [t.cpu().numpy() for t in code.decode_ms_batch(d, 25, variant=BS | 256)]
        for rep in range(2):
            got = [t.cpu().numpy() for t in code.decode_ms_batch(d, 25, variant=BS)]
            assert all((a == b).all() for a, b in zip(ref, got)), (frames, rep)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            got = code.decode_ms_batch(d, 25, variant=BS)
        st.synchronize()
        assert all((a == b.cpu().numpy()).all() for a, b in zip(ref, got)), frames
    # a large batch against the default dispatch (which IS the refill kernel at this size) and the oracle on a sample
    big = torch.from_numpy(np.tile(llrs, (44, 1))).cuda()                 # 67 584 frames (the default dispatch refills from 65 536)
    import labrador_ldpc_amd as la
    assert la.lib.labrador_ldpc_hip_decode_ms_i8_kernel(int(code), 0, big.shape[0]).decode().endswith("refill_kernel")
    a = code.decode_ms_batch(big, 25)
    b_ = code.decode_ms_batch(big, 25, variant=BS | 256)
    assert all(bool((x == y).all()) for x, y in zip(a, b_))
    oc, ic, kc, _ = oracle.decode_ms_batch(code, llrs[:256], 25)
    assert (a[0][:256].cpu().numpy() == oc).all() and (a[1][:256].cpu().numpy() == ic).all() and (a[2][:256].cpu().numpy() == kc).all()


@pytest.mark.parametrize("code", [LDPCCode.TM1536, LDPCCode.TM1280], ids=lambda c: c.name)
def test_slot_refill_on_concurrent_streams(code):
    """Two host threads, a stream each, launching the slot-refill kernel back to back on batches of different sizes at the same time:
    every stream has a queue word of its own (claim_counter), which must be back at zero for the stream's next launch whatever the other
    stream is doing.  Results equal the lockstep kernel's, launch by launch."""
    import threading
    rng = np.random.default_rng(78)
    hi = 1.5 if code is LDPCCode.TM1280 else 0.0
    llrs = np.concatenate([oracle.awgn_llrs(code, rng, 700, e, np.int8, scale=8.0, lim=31)[0] for e in (3.0 + hi, 1.0, 5.0 + hi)])
    rng.shuffle(llrs)
    dev_llrs = torch.from_numpy(np.tile(llrs, (10, 1))).cuda()                      # 21 000 frames: more than the resident slots
    sizes = (21000, 17, 4099, 1, 20999, 333)
    ref = {n: [t.cpu().numpy() for t in code.decode_ms_batch(dev_llrs[:n], 25, variant=BS | 256)] for n in sizes}
    torch.cuda.synchronize()
    errors = []

    def worker(order):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(6):
                    for n in order:
                        got = code.decode_ms_batch(dev_llrs[:n], 25, variant=BS)
                        st.synchronize()
                        if not all((a == b.cpu().numpy()).all() for a, b in zip(ref[n], got)):
                            errors.append((n, rep))
        except Exception as e:                                                        # noqa: BLE001 -- reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(o,)) for o in (sizes, sizes[::-1])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:4]
