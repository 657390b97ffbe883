"""The bit-sliced i8 decoder (labrador_ldpc_amd/csrc/decode_ms_bitslice.hpp) validated WITHOUT a GPU: tests/c/bitslice_emu.cpp
instantiates the kernel's own source text with a backend whose wave register is an array of 64 lanes (ds_bpermute, LDS and global
accesses as plain loops) and this file compares its results -- hard bits, iteration counts, success flags -- with the CPU oracle and
with the frozen i8 golden files, for the six TM codes (the layout is built on their quarter-wise permutations pi_k).  What it
pins: the index -> (lane, bit) layout and its lane permutations / word rotations, the plane arithmetic (saturating add / sub,
the sign-split key of |v|, the two running minima), the compressed row state, the LLR transposition and the output packing,
codewords of one wave finishing in different iterations.  The GPU tests then only have to show that gfx950 executes the same
text the same way (tests/test_gpu_bitslice.py)."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TM = ["TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]


@pytest.fixture(scope="module")
def emu():
    """decode(code, llrs, maxiters) through the emulated kernel; one shared object per code, the stale ones rebuilt in parallel
    (fully unrolled 64-lane code: ~1 minute each at -O1)."""
    src = [os.path.join(ROOT, "tests", "c", "bitslice_emu.cpp"), os.path.join(ROOT, "labrador_ldpc_amd", "csrc", "decode_ms_bitslice.hpp"),
           os.path.join(ROOT, "labrador_ldpc_amd", "csrc", "decode_ms_bitslice_split.hpp"), os.path.join(ROOT, "labrador_ldpc_amd", "csrc", "decode_bf_bitslice.hpp"),
           os.path.join(ROOT, "labrador_ldpc_amd", "csrc", "codes.hpp")]
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    libs, jobs = {}, []
    for name in TM:
        lib = os.path.join(ROOT, "build", f"libbitslice_emu_{name}.so")
        libs[name] = lib
        if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src):
            jobs.append(subprocess.Popen(["g++", "-O1", "-std=c++20", "-shared", "-fPIC", f"-DEMU_CODE={name}",
                                          "-I" + os.path.join(ROOT, "labrador_ldpc_amd", "csrc"), src[0], "-o", lib]))
    assert all(j.wait() == 0 for j in jobs)
    loaded = {}
    for name, lib in libs.items():
        L = ctypes.CDLL(lib)
        L.bs_emu_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        L.bs_emu_decode_bf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        L.bs_emu_decode_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        L.bs_emu_decode_refill.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        L.bs_emu_decode_split_refill.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        assert L.bs_emu_code() == oracle.CODES.index(name)
        loaded[oracle.CODES.index(name)] = L

    def decode(code, llrs, maxiters, split=False, refill=False):
        """split: the two-waves-per-group kernel of the rate-4/5 codes (decode_ms_bitslice_split.hpp), its two halves run alternately;
        refill: the slot-refill driver (decode_refill; with split: decode_refill_split, the two halves drawing the same chunks of 5
        frames each with a cursor of its own): ONE emulated wave (pair) takes the whole batch, a finished slot the next frame"""
        llrs = np.ascontiguousarray(llrs, dtype=np.int8)
        B = llrs.shape[0]
        out = np.full((B, oracle.output_len(code)), 0xEE, np.uint8)
        it, ok = np.full(B, 0xEEEEEEEE, np.uint32), np.full(B, 0xEE, np.uint8)
        L = loaded[code]
        fn = (L.bs_emu_decode_split_refill if split else L.bs_emu_decode_refill) if refill else L.bs_emu_decode_split if split else L.bs_emu_decode
        assert fn(llrs.ctypes.data, out.ctypes.data, it.ctypes.data, ok.ctypes.data, B, maxiters) == 0
        return out, it, ok
    def decode_bf(code, hard, maxiters):
        hard = np.ascontiguousarray(hard, dtype=np.uint8)
        B = hard.shape[0]
        out = np.full((B, oracle.output_len(code)), 0xEE, np.uint8)
        it, ok = np.full(B, 0xEEEEEEEE, np.uint32), np.full(B, 0xEE, np.uint8)
        assert loaded[code].bs_emu_decode_bf(hard.ctypes.data, out.ctypes.data, it.ctypes.data, ok.ctypes.data, B, maxiters) == 0
        return out, it, ok
    decode.group = lambda code: loaded[code].bs_emu_group()
    decode.bf = decode_bf
    return decode


def _same(emu, code, llrs, maxiters, split=False, refill=False):
    o, i, k = emu(code, llrs, maxiters, split, refill)
    oc, ic, kc, _ = oracle.decode_ms_batch(code, llrs, maxiters)
    bad = np.nonzero((o != oc).any(axis=1) | (i != ic) | (k != kc))[0]
    assert bad.size == 0, f"frames {bad.tolist()[:8]} differ (iters {i[bad][:8].tolist()} vs {ic[bad][:8].tolist()})"
    return i, k


@pytest.mark.parametrize("name", TM)
def test_emulated_kernel_equals_the_oracle_on_awgn_frames(emu, name):
    code = oracle.CODES.index(name)
    rng = np.random.default_rng(100 + code)
    # converging early / late / never, at the usual scale and at a saturating one; odd frame counts leave the last wave part-filled
    for ebn0, scale, lim, frames in ((3.5 if name in ("TM1280", "TM5120") else 2.5, 8.0, 31, 2 * emu.group(code) + 1), (1.0, 30.0, 127, 3), (4.5, 16.0, 127, 5)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, np.int8, scale=scale, lim=lim)
        for maxiters in (0, 3, 25):
            it, ok = _same(emu, code, llrs, maxiters)
    assert ok.all() and len(set(it.tolist())) > 1        # (the last set at 25: everything converges, after different numbers of iterations)


@pytest.mark.parametrize("name", ["TM1280", "TM5120"])
def test_emulated_two_wave_kernel_equals_the_oracle(emu, name):
    """The rate-4/5 codes' default kernel: a codeword group shared by two waves, each owning a set of block columns with all their
    edges (decode_ms_bitslice_split.hpp).  The two halves run stage by stage on one LDS store -- the order the workgroup barriers
    enforce on the GPU; the runner fails if they ever disagree on a verdict.  Pins: the column partition, the exchange of the partial
    row states and their merge (two smallest keys of a union, the arg-min slot, sign and parity), the unshared row's parity crossing
    over, corner inputs with ties everywhere."""
    code = oracle.CODES.index(name)
    rng = np.random.default_rng(300 + code)
    for ebn0, scale, lim, frames in ((3.5, 8.0, 31, 2 * emu.group(code) + 1), (1.0, 30.0, 127, 3), (4.5, 16.0, 127, 5), (4.0, 8.0, 31, emu.group(code) + 2)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, np.int8, scale=scale, lim=lim)
        for maxiters in (0, 3, 25):
            it, ok = _same(emu, code, llrs, maxiters, split=True)
    assert ok.all() and len(set(it.tolist())) > 1
    N = oracle.CODE_N[code] if hasattr(oracle, "CODE_N") else llrs.shape[1]
    corner = np.stack([np.zeros(N, np.int8), np.full(N, -128, np.int8), np.full(N, 127, np.int8), rng.integers(-128, 128, N).astype(np.int8),
                       rng.choice(np.array([-128, 127, 0, 1, -1], np.int8), N)])
    for maxiters in (25, 1, 2):
        _same(emu, code, corner, maxiters, split=True)


@pytest.mark.parametrize("name", ["TM1536", "TM2048", "TM6144"])
def test_emulated_slot_refill_equals_the_oracle(emu, name):
    """Slot refill (decode_refill, round 6): a slot whose codeword is finished hands in its results and takes the wave's next frame
    while the other slots iterate.  Frames that finish after 3 ... 25 iterations, frames that never do and saturating ones, shuffled, so
    that slots finish in every order; batches smaller than a wave's slots; every iteration cap that makes fresh slots expire at once or
    in step.  Pins: the cooperative per-slot prologue (one (block column, lane) unit per lane; TM2048: the codeword passes the staging
    slab in two pieces, behind the "v != 0" planes of the active slots) and epilogue, the masked state reset, the look-ahead frame."""
    code = oracle.CODES.index(name)
    rng = np.random.default_rng(900 + code)
    G = emu.group(code)
    parts = [oracle.awgn_llrs(code, rng, f, e, np.int8, scale=s, lim=l)[0] for e, s, l, f in
             ((2.5, 8.0, 31, 3 * G + 1), (1.0, 30.0, 127, 3), (4.5, 16.0, 127, G + 2), (2.0, 8.0, 31, G))]
    llrs = np.concatenate(parts)
    rng.shuffle(llrs)
    for frames in (llrs.shape[0], 1, max(1, G - 1)):
        for maxiters in (0, 1, 2, 7, 25):
            it, ok = _same(emu, code, llrs[:frames], maxiters, refill=True)
    N = llrs.shape[1]
    corner = np.stack([np.zeros(N, np.int8), np.full(N, -128, np.int8), np.full(N, 127, np.int8), rng.integers(-128, 128, N).astype(np.int8),
                       rng.choice(np.array([-128, 127, 0, 1, -1], np.int8), N)])
    for maxiters in (25, 1):
        _same(emu, code, corner, maxiters, refill=True)


@pytest.mark.parametrize("name", ["TM1280", "TM5120"])
def test_emulated_two_wave_slot_refill_equals_the_oracle(emu, name):
    """Slot refill on the two-wave kernel (decode_refill_split): both waves of a group keep the slots' iteration counts and frame
    numbers in a register each (lane = slot's lane), reach the same verdicts from the exchanged row states, and draw the same chunks
    of the frame supply; each places only the block columns it owns.  Pins: the vector bookkeeping, the owned-column literals of the
    prologue and epilogue, the chunked supply running dry in the middle of an event, the look-ahead frame across a chunk border."""
    code = oracle.CODES.index(name)
    rng = np.random.default_rng(950 + code)
    G = emu.group(code)
    parts = [oracle.awgn_llrs(code, rng, f, e, np.int8, scale=s, lim=l)[0] for e, s, l, f in
             ((3.5, 8.0, 31, 3 * G + 1), (1.0, 30.0, 127, 3), (5.5, 16.0, 127, G + 2), (3.0, 8.0, 31, G))]
    llrs = np.concatenate(parts)
    rng.shuffle(llrs)
    for frames in (llrs.shape[0], 1, max(1, G - 1), 5, 6):
        for maxiters in (0, 1, 2, 7, 25):
            it, ok = _same(emu, code, llrs[:frames], maxiters, split=True, refill=True)
    N = llrs.shape[1]
    corner = np.stack([np.zeros(N, np.int8), np.full(N, -128, np.int8), np.full(N, 127, np.int8), rng.integers(-128, 128, N).astype(np.int8),
                       rng.choice(np.array([-128, 127, 0, 1, -1], np.int8), N)])
    for maxiters in (25, 1):
        _same(emu, code, corner, maxiters, split=True, refill=True)


@pytest.mark.parametrize("name", ["TM2048", "TM8192"])
def test_sixteen_plane_build_equals_the_i16_oracle(name):
    """Round 4's review, item 4 (i): the plane count is a parameter of the kernel text (BS_PLANES, decode_ms_bitslice.hpp).  Built with
    16 planes the same text decodes with i16 saturation: sign-extended i8 LLRs through the i8 loader must give exactly what the oracle's
    decode_ms::<i16> (src/decoder.rs:51-59) gives on those values as i16 -- including frames whose marginals leave the i8 range (scale 30:
    sums of six +-127 messages), where the 8-plane build saturates and the 16-plane one must not.  (The GPU rate of this build:
    profiles/r05_kbench/bs_i16_tc.txt.)"""
    code = oracle.CODES.index(name)
    lib = os.path.join(ROOT, "build", f"libbitslice_emu16_{name}.so")
    src = [os.path.join(ROOT, "tests", "c", "bitslice_emu.cpp")] + [os.path.join(ROOT, "labrador_ldpc_amd", "csrc", f) for f in
           ("decode_ms_bitslice.hpp", "decode_ms_bitslice_split.hpp", "decode_bf_bitslice.hpp", "codes.hpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in src):
        subprocess.check_call(["g++", "-O1", "-std=c++20", "-shared", "-fPIC", f"-DEMU_CODE={name}", "-DBS_PLANES=16",
                               "-I" + os.path.join(ROOT, "labrador_ldpc_amd", "csrc"), src[0], "-o", lib])
    L = ctypes.CDLL(lib)
    L.bs_emu_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
    rng = np.random.default_rng(1600 + code)
    differs_from_i8 = False
    for ebn0, scale, lim, frames in ((2.5, 8.0, 31, 5), (1.0, 30.0, 127, 3), (4.5, 30.0, 127, 4), (3.0, 60.0, 127, 3)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, np.int8, scale=scale, lim=lim)
        if scale > 8:
            llrs[:, ::11] = -128                                    # |-128| = 128 in i16: no saturating abs there
        for maxiters in (25, 3, 0):
            B = llrs.shape[0]
            out = np.full((B, oracle.output_len(code)), 0xEE, np.uint8)
            it, ok = np.full(B, 0xEEEEEEEE, np.uint32), np.full(B, 0xEE, np.uint8)
            assert L.bs_emu_decode(np.ascontiguousarray(llrs).ctypes.data, out.ctypes.data, it.ctypes.data, ok.ctypes.data, B, maxiters) == 0
            oc, ic, kc, _ = oracle.decode_ms_batch(code, llrs.astype(np.int16), maxiters)
            assert (out == oc).all() and (it == ic).all() and (ok == kc).all(), (name, ebn0, scale, maxiters)
            o8, i8_, k8, _ = oracle.decode_ms_batch(code, llrs, maxiters)
            differs_from_i8 = differs_from_i8 or bool((o8 != oc).any() or (i8_ != ic).any())
    assert differs_from_i8                                           # (the inputs do exercise the difference between the two types)


@pytest.mark.parametrize("name", ["TM1280", "TM2048", "TM6144"])
def test_emulated_kernel_on_corner_inputs(emu, name):
    code = oracle.CODES.index(name)
    N = oracle.n(code)
    rng = np.random.default_rng(11)
    frames = [np.zeros(N, np.int8), np.full(N, -128, np.int8), np.full(N, 127, np.int8), rng.integers(-128, 128, N).astype(np.int8),
              rng.choice(np.array([-128, 127, 0, 1, -1], np.int8), N)]
    base, _ = oracle.awgn_llrs(code, rng, 3, 3.0, np.int8, scale=60.0, lim=127)
    spiked = base.copy()
    spiked[:, ::7] = -128                                   # |-128| = 127 (decoder.rs:46) inside otherwise decodable frames
    llrs = np.stack(frames + list(spiked) + list(base))
    for maxiters in (25, 1, 2, 60):
        _same(emu, code, llrs, maxiters)


def test_emulated_kernel_reproduces_the_i8_golden_files(emu):
    """tests/golden/awgn_<code>_i8.npz: frozen frames with the results both restatements agreed on -- no oracle in this loop."""
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "awgn_TM*_i8.npz")))
    assert len(files) == 6
    for f in files:
        z = np.load(f)
        code = oracle.CODES.index(os.path.basename(f).split("_")[1])
        for maxiters in (25, 4, 0):
            o, i, k = emu(code, z["llrs"], maxiters)
            assert (o == z[f"output_{maxiters}"]).all() and (i == z[f"iters_{maxiters}"]).all() and (k == z[f"success_{maxiters}"]).all(), (f, maxiters)


@pytest.mark.parametrize("name", TM)
def test_emulated_bit_sliced_decode_bf_equals_the_oracle(emu, name):
    """csrc/decode_bf_bitslice.hpp (decode_bf + the erasure pre-pass, decoder.rs:144-301) lane by lane against the oracle's restatement:
    clean codewords, the reference's three-flip scenario (decoder.rs:647-670), random error patterns that converge slowly or fail,
    part-filled waves, max_iters 0 / 1 / 2 / 20."""
    code = oracle.CODES.index(name)
    rng = np.random.default_rng(300 + code)
    N, K = oracle.n(code), oracle.k(code)
    B = 2 * emu.group(code) + 3
    hard = np.zeros((B, N // 8), dtype=np.uint8)
    for f in range(B):
        cw = oracle.copy_encode(code, rng.integers(0, 256, K // 8, dtype=np.uint8))
        nerr = 0 if f == 0 else (3 if f == 1 else int(rng.integers(0, max(2, N // 40))))
        if f == 1:
            cw[0] ^= 0xA8
        else:
            for pos in rng.choice(N, nerr, replace=False):
                cw[pos // 8] ^= 1 << (7 - pos % 8)
        hard[f] = cw
    seen_ok = False
    for maxiters in (20, 0, 1, 2):
        o, i, k = emu.bf(code, hard, maxiters)
        for f in range(B):
            ok_c, it_c, out_c = oracle.decode_bf(code, hard[f], maxiters)
            assert (bool(k[f]), int(i[f])) == (ok_c, it_c), (name, maxiters, f, int(i[f]), it_c)
            assert (o[f] == out_c).all(), (name, maxiters, f)
        seen_ok = seen_ok or bool(k.any())
    assert seen_ok
