"""Wider seeded sweep of the TM8192 default kernel (pair ownership, clamp-free f32 path, 64-bit odd-rotation
reads for i8/i16) and of TM2048 / TM6144 against the oracle: several seeds and operating points across the
waterfall, a few thousand frames in total.  Sized for ~20 s on the GPU box (the oracle is the slow side)."""
import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode

pytestmark = pytest.mark.gpu


def _check(code, llrs, maxiters):
    out_g, it_g, ok_g = code.decode_ms_batch(llrs, maxiters)
    out_c, it_c, ok_c, _ = oracle.decode_ms_batch(code, llrs, maxiters)
    bad = np.nonzero((it_g != it_c) | (ok_g != ok_c) | (out_g != out_c).any(axis=1))[0]
    assert bad.size == 0, f"{code.name} {llrs.dtype}: {bad.size}/{len(llrs)} frames differ, first {bad[0]}"
    return it_c, ok_c


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16], ids=["f32", "i8", "i16"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tm8192_sweep(dtype, seed):
    code = LDPCCode.TM8192
    rng = np.random.default_rng(1000 * seed + np.dtype(dtype).itemsize)
    conv = []
    for ebn0, maxiters in ((0.8, 10), (1.3, 25), (1.6, 40), (2.0, 25), (3.0, 25), (6.0, 12)):
        scale, lim = (8.0, 31) if dtype == np.int8 else (64.0, 4095)
        llrs, _ = oracle.awgn_llrs(code, rng, 72, ebn0, dtype, scale=scale, lim=lim)
        it, ok = _check(code, llrs, maxiters)
        conv.append(ok.mean())
    assert conv[0] < 0.5 and conv[-1] == 1.0          # both sides of the waterfall were exercised


@pytest.mark.parametrize("code", [LDPCCode.TM2048, LDPCCode.TM6144], ids=lambda c: c.name)
def test_other_codes_sweep(code):
    rng = np.random.default_rng(77 + int(code))
    for ebn0 in (1.0, 1.8, 2.6, 4.0):
        llrs, _ = oracle.awgn_llrs(code, rng, 200, ebn0, np.float32)
        _check(code, llrs, 30)
