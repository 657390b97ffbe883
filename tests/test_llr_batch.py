"""Batched LLR helpers (labrador_ldpc_hard_to_llrs_batch_* / labrador_ldpc_llrs_to_hard_batch_*): src/decoder.rs:484-509
frame after frame.  Host buffers are converted by the library's host code (CPU tests), device buffers by the
streaming kernels of csrc/llr_convert.hip (GPU tests).  The oracle's own restatement of the two functions and a numpy
statement of their definition are the references; the reference's vectors (src/decoder.rs:553-605) are replayed
through the batched forms as well."""
import json
import os

import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode, LdpcHipError

DTYPES = {"i8": np.int8, "i16": np.int16, "i32": np.int32, "f32": np.float32, "f64": np.float64}
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _bits(code, frames, seed):
    return np.random.default_rng(seed).integers(0, 256, (frames, code.n() // 8), dtype=np.uint8)


def _expected_llrs(bits, dtype):
    return np.where(np.unpackbits(bits, axis=1) == 1, -1, 1).astype(dtype)


def _corner_llrs(code, frames, dtype, seed):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        x = rng.integers(info.min, info.max, (frames, code.n()), dtype=dtype, endpoint=True)
        x[0, ::3] = 0
        x[1 % frames, :] = info.min
        x[2 % frames, ::2] = -1
    else:
        x = rng.normal(0, 3, (frames, code.n())).astype(dtype)
        x[0, ::3] = 0.0
        x[0, 1::3] = -0.0                  # -0.0 < 0 is false: a clear bit (decoder.rs:504)
        x[1 % frames, ::5] = np.nan        # NaN < 0 is false
        x[2 % frames, ::7] = -np.inf
        x[2 % frames, 1::7] = np.finfo(dtype).tiny * dtype(-0.5)      # negative denormal: a set bit
    return x


@pytest.mark.parametrize("name", ["TC128", "TM1280", "TM8192"])
@pytest.mark.parametrize("dt", list(DTYPES))
def test_host_batches_equal_the_definition_and_the_oracle(name, dt):
    code = LDPCCode[name]
    bits = _bits(code, 7, 5)
    llrs = code.hard_to_llrs_batch(bits, dt)
    assert llrs.dtype == DTYPES[dt] and (llrs == _expected_llrs(bits, DTYPES[dt])).all()
    for f in range(bits.shape[0]):
        assert (llrs[f] == oracle.hard_to_llrs(code, bits[f], DTYPES[dt])).all()
    x = _corner_llrs(code, 7, DTYPES[dt], 9)
    hard = code.llrs_to_hard_batch(x)
    assert (hard == np.packbits(x < 0, axis=1)).all()
    for f in range(x.shape[0]):
        assert (hard[f] == oracle.llrs_to_hard(code, x[f])).all()
    assert (code.llrs_to_hard_batch(llrs) == bits).all()                      # round trip


def test_reference_vectors_through_the_batched_forms():
    """src/decoder.rs:553-605: the reference's own hard_to_llrs / llrs_to_hard vectors (TC128)."""
    kat = KATS["hard_llr"]
    code = LDPCCode[kat["code"]]
    bits = np.array(kat["hard"], dtype=np.uint8)[None, :]
    want = np.array(kat["llrs"], dtype=np.int8)[None, :]
    assert (code.hard_to_llrs_batch(bits, "i8") == want).all()
    assert (code.llrs_to_hard_batch(want) == bits).all()


def test_argument_checks():
    code = LDPCCode.TC256
    with pytest.raises(ValueError):
        code.hard_to_llrs_batch(np.zeros((2, 5), dtype=np.uint8))
    with pytest.raises(ValueError):
        code.llrs_to_hard_batch(np.zeros((2, code.n() + 1), dtype=np.float32))
    with pytest.raises(ValueError):
        code.llrs_to_hard_batch(np.zeros((2, code.n()), dtype=np.float16))
    with pytest.raises(ValueError):
        code.hard_to_llrs_batch(np.zeros((2, code.n() // 8), dtype=np.uint8), "f32", llrs=np.zeros((2, code.n()), dtype=np.float64))
    assert code.hard_to_llrs_batch(np.zeros((0, code.n() // 8), dtype=np.uint8)).shape == (0, code.n())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["TC128", "TM2048", "TM8192"])
@pytest.mark.parametrize("dt", list(DTYPES))
def test_device_batches(name, dt):
    torch = pytest.importorskip("torch")
    code = LDPCCode[name]
    dev = torch.device("cuda", 0)
    for frames in (1, 3, 1000):
        bits = _bits(code, frames, 11 + frames)
        d_llrs = code.hard_to_llrs_batch(torch.from_numpy(bits).to(dev), dt)
        assert (d_llrs.cpu().numpy() == _expected_llrs(bits, DTYPES[dt])).all()
        x = _corner_llrs(code, frames, DTYPES[dt], 13 + frames)
        d_hard = code.llrs_to_hard_batch(torch.from_numpy(x).to(dev))
        assert (d_hard.cpu().numpy() == np.packbits(x < 0, axis=1)).all()
        assert torch.equal(code.llrs_to_hard_batch(d_llrs), torch.from_numpy(bits).to(dev))


@pytest.mark.gpu
def test_device_chain_encode_to_llrs_to_decode():
    """encode_batch -> hard_to_llrs_batch -> decode_ms_batch -> llrs of the output -> the same codewords, all
    device-resident: the data formats either side of the path without a host round trip."""
    torch = pytest.importorskip("torch")
    code = LDPCCode.TM2048
    dev = torch.device("cuda", 0)
    data = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (513, code.k() // 8), dtype=np.uint8)).to(dev)
    cw = code.encode_batch(data)
    llrs = code.hard_to_llrs_batch(cw, "i8")
    out, iters, ok = code.decode_ms_batch(llrs, 10)
    assert bool((ok == 1).all()) and bool((iters <= 1).all())
    assert torch.equal(out[:, : code.n() // 8], cw)
    assert torch.equal(code.llrs_to_hard_batch(llrs), cw)


@pytest.mark.gpu
def test_misaligned_device_llrs_are_refused():
    torch = pytest.importorskip("torch")
    code = LDPCCode.TC128
    dev = torch.device("cuda", 0)
    flat = torch.zeros(2 * code.n() + 1, dtype=torch.float32, device=dev)
    view = flat[1:].view(2, code.n())                                         # 4-byte aligned only
    with pytest.raises(LdpcHipError):
        code.llrs_to_hard_batch(view)
