"""Batched GPU bit-flipping decoder (labrador_ldpc_decode_bf[_batch]) against the CPU oracle, bit for bit:
output bytes (incl. the punctured bits the erasure pre-pass reconstructs), iterations and success."""
import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode

pytestmark = pytest.mark.gpu


def _compare(code, hard, maxiters):
    out_g, it_g, ok_g = code.decode_bf_batch(hard, maxiters)
    for f in range(len(hard)):
        ok_c, it_c, out_c = oracle.decode_bf(code, hard[f], maxiters)
        assert (bool(ok_g[f]), int(it_g[f])) == (ok_c, it_c), f"{code.name} frame {f}"
        assert (out_g[f] == out_c).all(), f"{code.name} frame {f}"
    return ok_g


@pytest.mark.parametrize("code", list(LDPCCode), ids=lambda c: c.name)
def test_three_flip_scenario(code):
    """test_decode_bf of the reference (src/decoder.rs:647-670)."""
    cw = oracle.copy_encode(code, np.arange(code.k() // 8, dtype=np.uint8))
    rx = cw.copy()
    rx[0] ^= 0xA8
    out = np.zeros(code.output_len(), dtype=np.uint8)
    ok, iters = code.decode_bf(rx, out, maxiters=50)
    ok_c, it_c, out_c = oracle.decode_bf(code, rx, 50)
    assert ok and ok_c and iters == it_c and (out == out_c).all()
    assert (out[: code.n() // 8] == cw).all()


@pytest.mark.parametrize("code", list(LDPCCode), ids=lambda c: c.name)
def test_random_error_patterns(code):
    """0 .. many random bit errors: converging, slowly converging and failing frames."""
    rng = np.random.default_rng(300 + int(code))
    B = 96
    hard = np.zeros((B, code.n() // 8), dtype=np.uint8)
    for f in range(B):
        cw = oracle.copy_encode(code, rng.integers(0, 256, code.k() // 8, dtype=np.uint8))
        nerr = int(rng.integers(0, max(2, code.n() // 40)))
        for pos in rng.choice(code.n(), nerr, replace=False):
            cw[pos // 8] ^= 1 << (7 - pos % 8)
        hard[f] = cw
    ok = _compare(code, hard, 20)
    assert ok.any()


@pytest.mark.parametrize("maxiters", [0, 1, 2])
def test_small_maxiters(maxiters):
    rng = np.random.default_rng(9)
    for code in (LDPCCode.TC256, LDPCCode.TM1536):
        hard = rng.integers(0, 256, (16, code.n() // 8), dtype=np.uint8)       # garbage: never converges
        _compare(code, hard, maxiters)


TM_CODES = [c for c in LDPCCode if c.name.startswith("TM")]


@pytest.mark.parametrize("code", TM_CODES, ids=lambda c: c.name)
def test_bit_sliced_kernel_on_large_batches(code):
    """From 256 groups of 64 / (M/32) codewords up the TM codes run the bit-sliced kernel (csrc/decode_bf_bitslice.hpp: one register per
    block column, one wave per group).  It must equal the byte-per-variable kernel -- which the same frames reach in slices below the
    threshold -- and the oracle on a sample; an odd batch leaves the last wave part-filled."""
    import torch
    rng = np.random.default_rng(900 + int(code))
    G = 64 // (code.submatrix_size() // 32)
    B = 300 * G + 3
    pool = np.zeros((32, code.n() // 8), dtype=np.uint8)
    for i in range(32):
        pool[i] = oracle.copy_encode(code, rng.integers(0, 256, code.k() // 8, dtype=np.uint8))
    hard = pool[rng.integers(0, 32, B)].copy()
    nerr = rng.integers(0, max(2, code.n() // 50), B)
    for f in range(B):
        for pos in rng.choice(code.n(), int(nerr[f]), replace=False):
            hard[f, pos // 8] ^= 1 << (7 - pos % 8)
    for maxiters in (20, 0, 1):
        out, it, ok = code.decode_bf_batch(hard, maxiters)                      # host buffers
        d = torch.from_numpy(hard).cuda()
        out_d, it_d, ok_d = code.decode_bf_batch(d, maxiters)                   # device buffers
        torch.cuda.synchronize()
        assert (out_d.cpu().numpy() == out).all() and (it_d.cpu().numpy().astype(np.int64) == it.astype(np.int64)).all() and (ok_d.cpu().numpy() == ok).all()
        step = 50 * G                                                           # below the threshold: the byte-per-variable kernel
        for lo in range(0, B, step):
            o2, i2, k2 = code.decode_bf_batch(hard[lo:lo + step], maxiters)
            assert (o2 == out[lo:lo + step]).all() and (i2 == it[lo:lo + step]).all() and (k2 == ok[lo:lo + step]).all(), (code.name, maxiters, lo)
        for f in list(range(0, 40)) + [B - 1, B - 2]:
            ok_c, it_c, out_c = oracle.decode_bf(code, hard[f], maxiters)
            assert (bool(ok[f]), int(it[f])) == (ok_c, it_c) and (out[f] == out_c).all(), (code.name, maxiters, f)
    one = int(ok.sum())                                                         # (max_iters = 1: only the error-free frames succeed ...)
    out, it, ok = code.decode_bf_batch(hard, 20)
    assert 0 < one < int(ok.sum()) <= B                                         # ... at 20 most frames do
