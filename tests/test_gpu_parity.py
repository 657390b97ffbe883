"""GPU parity: the HIP decoder (through the C ABI) against the CPU oracle, bit for bit.

Bar (BASELINE.json north_star): decoded hard bits, iterations-to-converge and success flag equal
the reference CPU decode_ms exactly on identical inputs -- integer and float alike (tolerance 0)."""
import numpy as np
import pytest

import oracle
import labrador_ldpc_amd as la
from labrador_ldpc_amd import LDPCCode

pytestmark = pytest.mark.gpu

ALL = list(LDPCCode)


def _compare(code, llrs, maxiters, variant=0):
    out_g, it_g, ok_g = code.decode_ms_batch(llrs, maxiters, variant=variant)
    out_c, it_c, ok_c, _ = oracle.decode_ms_batch(code, llrs, maxiters)
    bad = np.nonzero((it_g != it_c) | (ok_g != ok_c) | (out_g != out_c).any(axis=1))[0]
    assert bad.size == 0, (f"{code.name} {llrs.dtype}: {bad.size}/{len(llrs)} frames differ, first {bad[0]}: "
                           f"gpu(it={it_g[bad[0]]},ok={ok_g[bad[0]]}) cpu(it={it_c[bad[0]]},ok={ok_c[bad[0]]})")
    return it_c, ok_c


@pytest.mark.parametrize("code", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16, np.int32, np.float64], ids=["f32", "i8", "i16", "i32", "f64"])
def test_three_flip_scenario(code, dtype):
    """test_decode_ms of the reference (src/decoder.rs:671-699): 3 flipped bits, +-1 LLRs, 50 iters."""
    cw = oracle.copy_encode(code, np.arange(code.k() // 8, dtype=np.uint8))
    rx = cw.copy()
    rx[0] ^= 0xA8
    llrs = oracle.hard_to_llrs(code, rx, dtype)
    out = np.zeros(code.output_len(), dtype=np.uint8)
    ok, iters = code.decode_ms(llrs, out, maxiters=50)
    ok_c, it_c, out_c = oracle.decode_ms(code, llrs, 50)
    assert ok and ok_c and iters == it_c
    assert (out[: code.n() // 8] == cw).all()
    assert (out == out_c).all()


@pytest.mark.parametrize("code", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16, np.float64], ids=["f32", "i8", "i16", "f64"])
def test_awgn_parity(code, dtype):
    """Seeded AWGN frames across the waterfall: early/late convergence and failures."""
    rng = np.random.default_rng(0x1DBC + int(code))
    frames = 96 if code.n() >= 5120 else 256
    for ebn0 in (0.5, 2.0, 3.5, 6.0):
        scale = 8.0 if dtype == np.int8 else 64.0
        lim = 31 if dtype == np.int8 else 4095
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0, dtype, scale=scale, lim=lim)
        it, ok = _compare(code, llrs, 25)
    assert ok.any()


@pytest.mark.parametrize("code", ALL, ids=lambda c: c.name)
def test_i32_parity(code):
    """decode_ms::<i32> (src/decoder.rs:60-68): genuine 32-bit saturating arithmetic on the GPU.  Scales from a few
    units (many exact ties and zeros) to LLRs at +-2^31 (every accumulation saturates; |INT_MIN| = INT_MAX, :64)."""
    rng = np.random.default_rng(0x132 + int(code))
    frames = 48 if code.n() >= 5120 else 128
    for ebn0, scale in ((2.5, 3.0), (2.0, 1000.0), (3.0, 2.0 ** 24 + 1), (2.5, 3e8), (2.0, 1.5e9), (1.0, 4e9)):
        llrs, _ = oracle.awgn_llrs(code, rng, frames, ebn0 + (1.5 if code.n() <= 1280 else 0.0), np.int32, scale=scale, lim=2 ** 31 - 1)
        if scale > 1e9:
            llrs[llrs == -(2 ** 31 - 1)] = -2 ** 31
        _compare(code, llrs, 25)
    llrs[:] = np.where(rng.random(llrs.shape) < 0.5, 2 ** 31 - 1, -2 ** 31).astype(np.int32)
    _compare(code, llrs, 6)
    for variant in ((2, 32) if code == LDPCCode.TM8192 else ()):
        _compare(code, llrs, 6, variant=variant)


@pytest.mark.parametrize("code", [LDPCCode.TC128, LDPCCode.TC512, LDPCCode.TM1280, LDPCCode.TM1536, LDPCCode.TM2048, LDPCCode.TM5120,
                                  LDPCCode.TM8192], ids=lambda c: c.name)
def test_saturating_i8(code):
    """Full-scale i8 LLRs (+-127, -128): saturating add/sub/abs paths (src/decoder.rs:42-50)."""
    rng = np.random.default_rng(7 + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 128, 3.0, np.int8, scale=100.0, lim=127)
    llrs[llrs == -127] = -128
    _compare(code, llrs, 20)


@pytest.mark.parametrize("code", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("dtype", [np.int8, np.int16], ids=["i8", "i16"])
def test_integer_self_correction_at_the_ends_of_the_range(code, dtype):
    """The narrow types' self-correction is v = fma(nv, clamp01(fma(nv, old, 1)), 0) (IntOps::self_correct, form 6): exact because
    an integer product is 0 or at least 1 in magnitude and never rounds across zero -- up to |nv| < 2^17 times |old| <= 2^15,
    above 2^24, where the product itself does round.  Frames that live at those ends: full-scale LLRs of random sign (every
    message saturates: i16 products around 2^31), full scale mixed with zeros and +-1 (the smallest nonzero messages against the
    largest), noisy frames scaled to clip at the type's limits, 50 iterations so that failing frames keep oscillating; every
    kernel variant of the code."""
    info = np.iinfo(dtype)
    rng = np.random.default_rng(0xF6 + int(code) + info.bits)
    frames = 24 if code.n() >= 5120 else 64
    n = code.n()
    full = np.where(rng.random((frames, n)) < 0.5, info.max, info.min).astype(dtype)
    mixed = full.copy()
    r = rng.random((frames, n))
    mixed[r < 0.25] = 0
    mixed[(r >= 0.25) & (r < 0.4)] = 1
    mixed[(r >= 0.4) & (r < 0.55)] = -1
    noisy, _ = oracle.awgn_llrs(code, rng, frames, 2.0 if code.n() > 1280 else 4.0, dtype, scale=info.max / 1.5, lim=info.max)
    noisy[noisy == -info.max] = info.min
    variants = {LDPCCode.TM8192: (0, 2), LDPCCode.TM1536: (0, 2), LDPCCode.TM6144: (0, 2)}.get(code, (0,))
    for llrs in (full, mixed, noisy):
        for variant in variants:
            _compare(code, llrs, 50, variant=variant)


@pytest.mark.parametrize("code", [LDPCCode.TC256, LDPCCode.TM1280, LDPCCode.TM1536, LDPCCode.TM5120, LDPCCode.TM8192], ids=lambda c: c.name)
def test_f32_corner_values(code):
    """Zeros, signed zeros, denormals, huge and infinite LLRs.  (TM1280 / TM1536 / TM5120 take the self-correction
    select through an integer borrow, Ops<float>::keep_unless_negative; TM5120 on the register-lean kernel.)"""
    rng = np.random.default_rng(11 + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 64, 3.0, np.float32)
    n = code.n()
    llrs[0, :] = 0.0
    llrs[1, :] = -0.0
    llrs[2, ::3] = 0.0
    llrs[3, ::5] = -0.0
    llrs[4] *= np.float32(1e-41)          # denormals
    llrs[5] *= np.float32(1e37)           # near overflow: sums reach inf
    llrs[6, ::7] = np.inf
    llrs[7, ::11] = -np.inf
    llrs[8] = np.where(rng.random(n) < 0.5, np.float32(3.4e38), np.float32(-3.4e38))
    # NaN LLRs: hard_bit is `x < 0.0` (src/decoder.rs:76), false for a NaN whatever its sign bit; quiet and signalling, both signs,
    # a whole frame of them, NaN next to infinities (tests/golden/nan_*.npz freeze the same for every code)
    qp, qn, sp, sn = np.array([0x7FC00000, 0xFFC00000, 0x7FA00000, 0xFFA00001], dtype=np.uint32).view(np.float32)
    u = llrs.view(np.uint32)
    for row, pats in ((9, [qp]), (10, [qn]), (11, [sp]), (12, [sn]), (13, [qn, qp, sn, sp])):
        for j, pos in enumerate(rng.permutation(n)[: 1 if row < 13 else 24]):
            u[row, pos] = np.array(pats[j % len(pats)]).view(np.uint32)
    u[14, :] = np.array(qn).view(np.uint32)
    u[15, ::2] = np.array(sn).view(np.uint32)
    llrs[16, ::13] = np.inf
    u[16, 5::13] = np.array(qn).view(np.uint32)
    assert np.isnan(llrs[9:17]).any(axis=1).all()
    _compare(code, llrs, 20)
    for variant in {LDPCCode.TM8192: (2, 32 + 256)}.get(code, (256,)):
        _compare(code, llrs, 20, variant=variant)


@pytest.mark.parametrize("code", [LDPCCode.TM5120, LDPCCode.TM1280], ids=lambda c: c.name)
def test_nan_two_pass(code):
    """The register-lean f32 kernels (TM5120, TM1280) handle NaN LLRs in TWO kernels unless the batch is small (decode_ms_launch.hpp,
    two_pass_nan()): the first decodes as if there were none and leaves a mark in `iters` for every codeword whose marginals
    show one, the second is the NaN-handling kernel over the marked codewords.  `variant` 512 / 1024 force one / two passes; the
    default decides by batch size.  All three must agree with the oracle frame for frame: NaNs of every kind (quiet, signalling,
    both signs, one per frame, whole frames, next to infinities) at the first, the last and runs of positions of a batch large
    enough for the default to take two passes; a batch without any; a batch of nothing else; no mark left behind."""
    import torch
    rng = np.random.default_rng(0x7A + int(code))
    n = code.n()
    clean, _ = oracle.awgn_llrs(code, rng, 24, 3.5, np.float32)
    clean[0, ::5] = -0.0
    clean[1, ::7] = np.inf
    clean[2, ::9] = -np.inf
    dirty, _ = oracle.awgn_llrs(code, rng, 12, 3.5, np.float32)
    qp, qn, sp, sn = (np.array([v], dtype=np.uint32) for v in (0x7FC00000, 0xFFC00000, 0x7FA00000, 0xFFA00001))
    u = dirty.view(np.uint32)
    for row, pat in enumerate((qp, qn, sp, sn)):
        u[row, rng.integers(n)] = pat[0]                       # one NaN in the frame
    for row in range(4, 8):
        for j, pos in enumerate(rng.permutation(n)[:40]):
            u[row, pos] = (qp, qn, sp, sn)[j % 4][0]
    u[8, :] = qn[0]
    u[9, ::2] = sn[0]
    dirty[10, ::13] = np.inf
    u[10, 5::13] = qn[0]
    u[11, n - 1] = qn[0]                                       # the very last LLR of a frame
    assert np.isnan(dirty).any(axis=1).all() and not np.isnan(clean).any()
    base = np.concatenate([clean, dirty])
    ref = oracle.decode_ms_batch(code, base, 25)
    nc, nd = len(clean), len(dirty)

    def check(idx, variants):
        d = torch.from_numpy(base[idx]).cuda()
        for variant in variants:
            out, it, ok = (t.cpu().numpy() for t in code.decode_ms_batch(d, 25, variant=variant))
            bad = np.nonzero((it != ref[1][idx]) | (ok != ref[2][idx]) | (out != ref[0][idx]).any(axis=1))[0]
            assert bad.size == 0, (code.name, len(idx), variant, bad[:8], it[bad[:8]], ref[1][idx][bad[:8]])

    check(np.arange(len(base)), (512, 1024, 1024 + 256, 0))                               # small batch, every kind of frame
    check(nc + np.arange(nd), (1024, 512))                                                # nothing but marked codewords
    check(np.arange(nc), (1024,))                                                         # no NaN anywhere
    check(np.array([nc + 1]), (1024, 512))                                                # batch of one
    big = 12000 if code == LDPCCode.TM5120 else 40000                                     # two passes by default
    idx = rng.integers(0, nc, big)
    idx[0] = nc                                                                           # first and last codeword of the launch
    idx[-1] = nc + 11
    idx[1000:1000 + nd] = nc + np.arange(nd)                                              # a run
    idx[rng.integers(0, big, 300)] = nc + rng.integers(0, nd, 300)                        # scattered
    check(idx, (0, 1024, 512, 0))
    check(rng.integers(0, nc, big), (0,))                                                 # large and clean
    d = torch.from_numpy(base[idx]).cuda()
    out, it, ok = (t.cpu().numpy() for t in code.decode_ms_batch(d, 0, variant=1024))     # zero iterations: one pass whatever was asked
    assert not out.any() and (it == 0).all() and (ok == 0).all()
    for maxiters in (1, 2):
        o_c, i_c, k_c, _ = oracle.decode_ms_batch(code, base, maxiters)
        out, it, ok = (t.cpu().numpy() for t in code.decode_ms_batch(torch.from_numpy(base).cuda(), maxiters, variant=1024))
        assert (out == o_c).all() and (it == i_c).all() and (ok == k_c).all(), maxiters


@pytest.mark.parametrize("code", [LDPCCode.TC128, LDPCCode.TC256, LDPCCode.TC512, LDPCCode.TM1280, LDPCCode.TM1536, LDPCCode.TM2048,
                                  LDPCCode.TM5120, LDPCCode.TM8192], ids=lambda c: c.name)
@pytest.mark.parametrize("maxiters", [0, 1, 2, 3])
def test_small_maxiters(code, maxiters):
    """Iteration caps around the early verdicts (the TC codes' in-wave verdict, TM1280 / TM1536's in-phase one): frames that
    are codewords already (iters 0), that converge at the cap, one short of it, or not at all."""
    rng = np.random.default_rng(5)
    llrs, _ = oracle.awgn_llrs(code, rng, 64, 4.0, np.float32)
    clean, _ = oracle.awgn_llrs(code, rng, 8, 30.0, np.float32)                # error-free: success at iteration 0
    it, ok = _compare(code, np.concatenate([llrs, clean]), maxiters)
    if maxiters >= 1 and code.punctured_bits() == 0:
        assert (ok[-8:] == 1).all() and (it[-8:] == 0).all()
    if maxiters >= 2:
        assert (ok[-8:] == 1).all() and (it[-8:] <= 1).all()                  # punctured codes need one iteration (decoder.rs:607-645)
    l8, _ = oracle.awgn_llrs(code, rng, 32, 5.0, np.int8)
    _compare(code, l8, maxiters)
    # the other LLR types, each after a longer decode of other frames has left its state in the LDS (the in-place f64 kernels
    # keep marginal sign words there: round 3 found them hard-deciding the PREVIOUS codeword's at max_iters 0)
    for dtype, scale, lim in ((np.float64, 1.0, 0), (np.int16, 64.0, 4095), (np.int32, 3e8, 2 ** 31 - 1)):
        other, _ = oracle.awgn_llrs(code, rng, 600, 1.0, dtype, scale=scale, lim=lim)
        code.decode_ms_batch(other, 6)
        lx, _ = oracle.awgn_llrs(code, rng, 600, 4.0, dtype, scale=scale, lim=lim)
        _compare(code, lx, maxiters)


def test_ragged_batches():
    """Batch sizes that do not fill a workgroup (TC128 packs 4 codewords per wave) and batch = 1."""
    rng = np.random.default_rng(9)
    for code in (LDPCCode.TC128, LDPCCode.TC256, LDPCCode.TM1280):
        llrs, _ = oracle.awgn_llrs(code, rng, 37, 3.0, np.float32)
        for b in (1, 2, 3, 5, 37):
            _compare(code, llrs[:b], 25)


def test_empty_batch():
    code = LDPCCode.TC128
    out, it, ok = code.decode_ms_batch(np.zeros((0, code.n()), dtype=np.float32), 25)
    assert out.shape == (0, code.output_len()) and it.shape == (0,) and ok.shape == (0,)


@pytest.mark.parametrize("code,variant", [(LDPCCode.TM8192, 2), (LDPCCode.TM8192, 4), (LDPCCode.TM8192, 32), (LDPCCode.TM2048, 2),
                                          (LDPCCode.TM2048, 32), (LDPCCode.TM1536, 2), (LDPCCode.TM6144, 2)],
                         ids=["TM8192-ipt2", "TM8192-ipt4", "TM8192-pair", "TM2048-ipt2", "TM2048-pair", "TM1536-ipt2", "TM6144-ipt2"])
def test_variants(code, variant):
    """Non-default kernels: (t, t + M/2) ownership with 2 or 4 indices per thread, pair ownership (2t, 2t + 1) = 32."""
    rng = np.random.default_rng(21)
    llrs, _ = oracle.awgn_llrs(code, rng, 96, 2.0, np.float32)
    llrs[0, ::7] = -0.0
    llrs[1] *= np.float32(1e37)
    _compare(code, llrs, 25, variant=variant)


@pytest.mark.parametrize("dtype", [np.int8, np.int16], ids=["i8", "i16"])
@pytest.mark.parametrize("variant", [2, 32])
def test_tm8192_integer_variants(dtype, variant):
    code = LDPCCode.TM8192
    rng = np.random.default_rng(23)
    llrs, _ = oracle.awgn_llrs(code, rng, 96, 2.5, dtype, scale=8.0 if dtype == np.int8 else 64.0, lim=31 if dtype == np.int8 else 4095)
    _compare(code, llrs, 25, variant=variant)


@pytest.mark.parametrize("code", [LDPCCode.TC128, LDPCCode.TM1280, LDPCCode.TM8192], ids=lambda c: c.name)
def test_f64_corner_values(code):
    """f64 path with signed zeros, denormals, infinities (real comparisons as in src/decoder.rs:78-86)."""
    rng = np.random.default_rng(31 + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 32, 3.0, np.float64)
    llrs[0, :] = 0.0
    llrs[1, :] = -0.0
    llrs[2, ::3] = -0.0
    llrs[3] *= 1e-310
    llrs[4] *= 1e307
    llrs[5, ::7] = np.inf
    llrs[6, ::11] = -np.inf
    # NaN LLRs (src/decoder.rs:85: `x < 0.0`): quiet / signalling, both signs, a whole frame
    qp, qn, sp, sn = np.array([0x7FF8000000000000, 0xFFF8000000000000, 0x7FF4000000000000, 0xFFF4000000000001], dtype=np.uint64).view(np.float64)
    u = llrs.view(np.uint64)
    for row, pats in ((7, [qp]), (8, [qn]), (9, [sp]), (10, [sn]), (11, [qn, qp, sn, sp])):
        for j, pos in enumerate(rng.permutation(code.n())[: 1 if row < 11 else 24]):
            u[row, pos] = np.array(pats[j % len(pats)]).view(np.uint64)
    u[12, :] = np.array(qn).view(np.uint64)
    assert np.isnan(llrs[7:13]).any(axis=1).all()
    _compare(code, llrs, 20)
    _compare(code, llrs, 20, variant=100)


F64_VARIANTS = [(LDPCCode.TC128, 17), (LDPCCode.TC256, 17), (LDPCCode.TC512, 17), (LDPCCode.TM1280, 1), (LDPCCode.TM1280, 33),
                (LDPCCode.TM1536, 17), (LDPCCode.TM2048, 1), (LDPCCode.TM2048, 33), (LDPCCode.TM5120, 18), (LDPCCode.TM5120, 33),
                (LDPCCode.TM6144, 33), (LDPCCode.TM6144, 18), (LDPCCode.TM6144, 2), (LDPCCode.TM8192, 36)]


@pytest.mark.parametrize("code,variant", F64_VARIANTS + [(c, 100) for c in ALL], ids=lambda v: getattr(v, "name", str(v)))
def test_f64_variants(code, variant):
    """f64: the non-default register-kernel instantiations (IPT = variant & 15, register-lean check phase if
    variant & 16, in-place messages if variant & 32) and the workspace kernel (variant 100) all agree with the oracle."""
    rng = np.random.default_rng(41 + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 48, 2.5, np.float64)
    llrs[0, ::5] = -0.0
    llrs[1] *= 1e-310
    _compare(code, llrs, 25, variant=variant)


def test_f64_unbuilt_variant_is_reported():
    code = LDPCCode.TM8192
    with pytest.raises(la.LdpcHipError, match="variant"):
        code.decode_ms_batch(np.zeros((2, code.n()), dtype=np.float64), 5, variant=1)   # 176 KB of LDS: not built


def test_tm8192_ragged_batches_on_the_pair_kernel():
    """Batch sizes around the persistent grid (256 workgroups on an MI355X): 1 frame, fewer frames than
    workgroups, one and a few more than the grid, so that workgroups decode 0, 1 or 2 codewords."""
    code = LDPCCode.TM8192
    rng = np.random.default_rng(77)
    base, _ = oracle.awgn_llrs(code, rng, 24, 1.6, np.float32)
    ref = oracle.decode_ms_batch(code, base, 25)
    for b in (1, 2, 255, 257, 300, 520):
        idx = np.arange(b) % len(base)
        out, it, ok = code.decode_ms_batch(base[idx], 25)
        assert (out == ref[0][idx]).all() and (it == ref[1][idx]).all() and (ok == ref[2][idx]).all(), b


def test_tm8192_clamp_mode_is_chosen_per_codeword():
    """The f32 pair kernel drops the FLT_MAX clamp of the exclusive minimum for codewords whose LLRs are all
    below a limit derived from max_iters (2^69 at 20 iterations) and keeps it for the others; a persistent workgroup decodes both kinds back to back
    (frames f, f + 256, f + 512 share a workgroup on an MI355X)."""
    code = LDPCCode.TM8192
    rng = np.random.default_rng(99)
    base, _ = oracle.awgn_llrs(code, rng, 40, 1.8, np.float32)
    llrs = base[np.arange(600) % 40].copy()
    for f in range(600):
        kind = (f // 7) % 4
        if kind == 1:
            llrs[f, (f * 13) % code.n()] = np.inf
        elif kind == 2:
            llrs[f] *= np.float32(1e37)                     # sums overflow to +-inf
        elif kind == 3:
            llrs[f, ::97] = np.float32(2.0 ** 64)           # large, but under the clamp-free path's limit at 20 iterations (2^69)
    _compare(code, llrs, 20)


def test_tm2048_clamp_mode_is_chosen_per_codeword():
    """Same for the (t)-ownership kernel, which uses the clamp-free check phase on TM2048 f32; 13 000 frames
    exceed the persistent grid (3 workgroups x 256 CUs x 16), so workgroups decode both kinds in sequence."""
    code = LDPCCode.TM2048
    rng = np.random.default_rng(98)
    base, _ = oracle.awgn_llrs(code, rng, 64, 2.2, np.float32)
    llrs = base[np.arange(13000) % 64].copy()
    for f in range(0, 13000, 3):
        kind = (f // 3) % 3
        if kind == 0:
            llrs[f, (f * 7) % code.n()] = -np.inf
        elif kind == 1:
            llrs[f] *= np.float32(1e37)
    _compare(code, llrs, 15)


@pytest.mark.parametrize("code", [LDPCCode.TM8192, LDPCCode.TM2048, LDPCCode.TC512, LDPCCode.TM1536], ids=lambda c: c.name)
def test_clamp_free_path_at_its_magnitude_limit(code):
    """the clamp-free check phase is exact only while no magnitude reaches FLT_MAX, and messages feed
    back, so the admissible |LLR| depends on the iteration count: the host passes 2^floor(126 - log2(7) max_iters)
    (nocap_limit_for, decode_ms_launch.hpp).  Frames with EVERY |LLR| at that limit, random signs, never
    converging, must agree with the oracle (whose minima start at FLT_MAX) at every max_iters -- on the
    clamp-free copy of the loop up to the limit, on the clamped copy just above it and beyond 45 iterations."""
    rng = np.random.default_rng(64)
    signs = np.where(rng.random((24, code.n())) < 0.5, 1.0, -1.0).astype(np.float32)
    for maxiters in (5, 20, 25, 28, 29, 44, 60, 300):
        e = int(np.floor(126.0 - 2.8074 * maxiters))
        # round 3: the TM8192 pair kernel's clamp form of the self-correction (v = med3(nv, 0, nv + old * 2^126)) narrows the
        # vote to 2^floor(82.5 - log2(7) max_iters) while that is >= 2^3 (max_iters <= 28) and runs the mul_legacy form beyond
        e2 = int(np.floor(82.5 - 2.8074 * maxiters))
        for mag in (2.0 ** max(e, -120), 2.0 ** min(max(e, -120) + 1, 127), 2.0 ** 64, 1.0,
                    2.0 ** max(e2, -19), 2.0 ** (max(e2, -19) + 1)):
            llrs = signs * np.float32(mag)
            it, ok = _compare(code, llrs, maxiters)
            assert (ok == 0).all() and (it == maxiters).all()
            llrs[:, ::3] *= np.float32(2.0 ** -20)                  # mixed magnitudes
            _compare(code, llrs, maxiters)


@pytest.mark.parametrize("code", [LDPCCode.TM8192, LDPCCode.TM2048, LDPCCode.TC512, LDPCCode.TM1536], ids=lambda c: c.name)
def test_bounded_mode_at_the_small_end_of_its_llr_range(code):
    """The clamp-free copy of the f32 loop also tests the self-correction by a multiply (nv * old < 0), which is
    exact only if no product underflows: its codewords must have every nonzero |LLR| >= 2^-20 (then every nonzero
    value of the decode is >= 2^-43).  Frames right at that bound take the multiply form, frames with one LLR
    below it (or denormal, or many exact zeros) the bit-operation form; all must equal the oracle."""
    rng = np.random.default_rng(20)
    y, _ = oracle.awgn_llrs(code, rng, 48, 2.0, np.float32)
    mags = (np.float32(2.0 ** -20) * (1.0 + rng.random(y.shape))).astype(np.float32)      # [2^-20, 2^-19)
    at_bound = np.copysign(mags, y).astype(np.float32)
    _compare(code, at_bound, 25)
    _compare(code, (y * np.float32(2.0 ** -18)).astype(np.float32), 25)                       # AWGN shape, many below the bound
    below = at_bound.copy()
    below[::2, 17] = np.float32(2.0 ** -21)                                                   # one LLR under the bound
    below[1::4, 5] = np.float32(1e-42)                                                        # one denormal
    below[3::4, ::9] = 0.0                                                                    # exact zeros are allowed in either mode
    _compare(code, below, 25)
    zeros = at_bound.copy()
    zeros[:, ::3] = 0.0
    zeros[:, 1::3] = -0.0
    _compare(code, zeros, 25)


@pytest.mark.parametrize("code", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("dtype", [np.float32, np.int8], ids=["f32", "i8"])
def test_queue_fed_and_fixed_stride_distribution_agree(code, dtype):
    """The workgroups of a launch take their codewords from the launch's queue (one atomic per chunk of codeword groups,
    decode_ms_kernel.hpp "dynamic distribution"; the reference's harness does the same with its workers,
    perftest/src/main.rs:39-45) or, with VARIANT_STATIC = 256, by a fixed stride.  Which workgroup decodes a frame must not
    matter: batch sizes around the chunk size and the resident grid, launches back to back on one stream (the queue head puts
    itself back to zero), and the oracle on the last one."""
    import torch
    rng = np.random.default_rng(0x51 + int(code))
    base, _ = oracle.awgn_llrs(code, rng, 40, {0: 4.0, 1: 3.5, 2: 3.0, 3: 3.6, 4: 2.6, 5: 2.0, 6: 3.3, 7: 2.3, 8: 1.9}[int(code)], dtype,
                               scale=8.0, lim=31)
    ref = oracle.decode_ms_batch(code, base, 25)
    sizes = (1, 7, 9, 255, 257, 1000, 3073, 6000) if code.n() <= 2048 else (1, 2, 255, 257, 511, 513, 800)
    for b in sizes:
        idx = (np.arange(b) * 7) % len(base)
        d = torch.from_numpy(base[idx]).cuda()
        for variant in (0, 256, 0, 0):
            out, it, ok = (t.cpu().numpy() for t in code.decode_ms_batch(d, 25, variant=variant))
            assert (out == ref[0][idx]).all() and (it == ref[1][idx]).all() and (ok == ref[2][idx]).all(), (b, variant)
    out, it, ok = (t.cpu().numpy() for t in code.decode_ms_batch(d, 0))          # max_iters 0: no queue (nothing to hide the draw behind)
    assert not out.any() and (it == 0).all() and (ok == 0).all()


def test_launches_on_different_streams_have_their_own_queue():
    """Launches of one stream run in order and share a queue head; launches of different streams may overlap and must not.
    Eight streams decode different batches at once, repeatedly; every frame must come out decoded exactly once."""
    import torch
    code = LDPCCode.TM2048
    rng = np.random.default_rng(0x57)
    base, _ = oracle.awgn_llrs(code, rng, 64, 2.2, np.float32)
    ref = oracle.decode_ms_batch(code, base, 25)
    streams = [torch.cuda.Stream() for _ in range(8)]
    jobs = []
    for rep in range(3):
        for si, st in enumerate(streams):
            b = 700 + 301 * si + rep
            idx = (np.arange(b) * (si + 3) + rep) % len(base)
            d = torch.from_numpy(base[idx]).cuda()
            out = torch.full((b, code.output_len()), 0xEE, dtype=torch.uint8, device="cuda")
            it = torch.full((b,), -1, dtype=torch.int32, device="cuda")
            ok = torch.full((b,), 7, dtype=torch.uint8, device="cuda")
            jobs.append((st, idx, d, out, it, ok))
    torch.cuda.synchronize()
    for st, idx, d, out, it, ok in jobs:
        with torch.cuda.stream(st):
            code.decode_ms_batch(d, 25, output=out, iters=it, success=ok, stream=st.cuda_stream)
    torch.cuda.synchronize()
    for st, idx, d, out, it, ok in jobs:
        assert (out.cpu().numpy() == ref[0][idx]).all() and (it.cpu().numpy() == ref[1][idx]).all() and (ok.cpu().numpy() == ref[2][idx]).all()


@pytest.mark.parametrize("code", [LDPCCode.TM8192, LDPCCode.TM2048, LDPCCode.TC512, LDPCCode.TM1536], ids=lambda c: c.name)
def test_clamp_form_of_the_self_correction_with_extreme_magnitude_ratios(code):
    """The clamp form needs big * |old| > |nv| for every nonzero old: frames that mix LLRs at the top of the admitted range
    with LLRs at its bottom (2^-20) and exact zeros -- so that tiny old messages meet huge new ones, with signs that disagree
    -- at iteration counts on both sides of the switch between the two forms, all never converging."""
    rng = np.random.default_rng(0xC1A)
    n = code.n()
    for maxiters in (8, 25, 28, 29, 40):
        top = 2.0 ** max(int(np.floor(82.5 - 2.8074 * maxiters)), 3)
        frames = []
        for f in range(16):
            mags = np.where(rng.random(n) < 0.5, top, 2.0 ** -20) * (1.0 + rng.random(n) * (f % 2))     # exact powers of two / dense mantissas
            x = np.where(rng.random(n) < 0.5, 1.0, -1.0) * mags
            x[rng.random(n) < 0.05] = 0.0
            frames.append(x)
        llrs = np.asarray(frames, dtype=np.float32)
        _compare(code, llrs, maxiters)
        _compare(code, (llrs * np.float32(2.0)).astype(np.float32), maxiters)           # one binade above: the clamped loop


def test_baseline_config1_through_the_single_frame_entry():
    """BASELINE.json configs[0] verbatim (TC128, one codeword, 50 iterations, AWGN 3 dB) through the reference-shaped
    labrador_ldpc_decode_ms_f32 (capi/src/lib.rs:113-119): the frame and its frozen result of tests/test_oracle_kats.py."""
    code = LDPCCode.TC128
    rng = np.random.default_rng(0x1DBC + int(code))
    llrs, _ = oracle.awgn_llrs(code, rng, 1, 3.0, np.float32)
    out = np.zeros(code.output_len(), dtype=np.uint8)
    ok, iters = code.decode_ms(llrs[0], out, maxiters=50)
    assert ok and iters == 3 and out.tobytes().hex() == "1883fafcd83eb9577b082a4694f949c4"
