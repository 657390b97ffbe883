"""Batched GPU encoder (labrador_ldpc_encode_batch) against the reference's parity known-answers
(src/encoder.rs:361-527) and the CPU oracle, bit for bit; plus encode -> channel -> decode round trips."""
import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("code", list(LDPCCode), ids=lambda c: c.name)
def test_encode_batch_matches_kat_and_oracle(code, kats):
    rng = np.random.default_rng(100 + int(code))
    B = 257                                           # not a multiple of anything
    data = rng.integers(0, 256, (B, code.k() // 8), dtype=np.uint8)
    data[0] = np.arange(code.k() // 8, dtype=np.uint8)              # the reference's KAT input
    data[1] = 0
    data[2] = 0xFF
    cw = code.encode_batch(data)
    assert cw.shape == (B, code.n() // 8)
    assert (cw[:, : code.k() // 8] == data).all()
    assert cw[0, code.k() // 8:].tolist() == kats["encode_parity"][code.name]
    assert not cw[1].any()                                           # the zero codeword
    for f in range(B):
        assert (cw[f] == oracle.copy_encode(code, data[f])).all(), f"frame {f}"


def test_encode_batch_device_tensors_and_round_trip():
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    for code, ebn0 in ((LDPCCode.TC512, 4.0), (LDPCCode.TM2048, 3.0), (LDPCCode.TM8192, 2.5)):
        B = 4096
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        data = torch.randint(0, 256, (B, code.k() // 8), dtype=torch.uint8, device=dev, generator=g)
        cw = code.encode_batch(data)
        torch.cuda.synchronize()
        # same as the host encoder
        h = cw[:16].cpu().numpy()
        for f in range(16):
            ref = np.zeros(code.n() // 8, dtype=np.uint8)
            code.copy_encode(data[f].cpu().numpy(), ref)
            assert (h[f] == ref).all()
        # encode -> AWGN -> decode recovers the data of every frame that reports success
        sigma = float(np.sqrt(1.0 / (2.0 * (code.k() / code.n()) * 10.0 ** (ebn0 / 10.0))))
        llrs = code.awgn_frames(cw, B, sigma, seed=77)                 # pool == batch: frame f <- codeword f
        out, iters, ok = code.decode_ms_batch(llrs, 50)
        torch.cuda.synchronize()
        okb = ok.bool()
        assert float(okb.float().mean()) > 0.95
        same = (out[:, : code.k() // 8] == data).all(dim=1)
        assert float(same[okb].float().mean()) > 0.999


@pytest.mark.parametrize("code", [LDPCCode.TM2048, LDPCCode.TM5120, LDPCCode.TM6144, LDPCCode.TM8192], ids=lambda c: c.name)
def test_encode_batch_resident_grid_every_workgroup(code):
    """A batch that fills the resident grid (the XCD-aware workgroup map of csrc/encode.hip is on: grid.y a multiple of 8) with a
    ragged last frame range: frames from every workgroup's range against the oracle, bit for bit."""
    rng = np.random.default_rng(300 + int(code))
    B = 4091
    data = rng.integers(0, 256, (B, code.k() // 8), dtype=np.uint8)
    cw = code.encode_batch(data)
    assert cw.shape == (B, code.n() // 8)
    assert (cw[:, : code.k() // 8] == data).all()
    for f in list(range(0, B, 5)) + list(range(B - 140, B)):
        assert (cw[f] == oracle.copy_encode(code, data[f])).all(), f"frame {f}"


def test_empty_and_bad_arguments():
    code = LDPCCode.TC128
    assert code.encode_batch(np.zeros((0, code.k() // 8), dtype=np.uint8)).shape == (0, code.n() // 8)
    with pytest.raises(ValueError):
        code.encode_batch(np.zeros((3, code.k() // 8 + 1), dtype=np.uint8))
