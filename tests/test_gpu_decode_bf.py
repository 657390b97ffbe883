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
