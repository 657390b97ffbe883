"""The CPU oracle against every known-answer the reference's own tests hold for this path
(tests/golden/reference_kats.json, extracted by tests/golden/make_reference_kats.py)."""
import numpy as np
import pytest

import oracle

CODES = list(range(9))


@pytest.mark.parametrize("code", CODES, ids=oracle.CODES)
def test_edge_order_crc(code, kats):
    """src/codes/mod.rs:517-535: edge count == paritycheck_sum and CRC-32 of the exact edge order."""
    name = oracle.CODES[code]
    chk, var = oracle.edges(code)
    assert len(chk) == kats["sizes"][name]["paritycheck_sum"]
    assert oracle.L.oracle_edge_crc(code) == kats["edge_crc"][code]
    # no duplicate edges, indices in range
    n, k, p = oracle.n(code), oracle.k(code), oracle.p(code)
    assert chk.max() == n + p - k - 1 and var.max() == n + p - 1
    assert len(set(zip(chk.tolist(), var.tolist()))) == len(chk)


@pytest.mark.parametrize("code", CODES, ids=oracle.CODES)
def test_size_tables(code, kats):
    """src/decoder.rs:531-551 against the CodeParams constants (src/codes/mod.rs:109-241)."""
    s = kats["sizes"][oracle.CODES[code]]
    L = oracle.L
    assert L.oracle_code_n(code) == s["n"] and L.oracle_code_k(code) == s["k"]
    assert L.oracle_code_punctured_bits(code) == s["punctured_bits"]
    assert L.oracle_code_submatrix_size(code) == s["submatrix_size"]
    assert L.oracle_code_circulant_size(code) == s["circulant_size"]
    assert L.oracle_bf_working_len(code) == s["decode_bf_working_len"]
    assert L.oracle_ms_working_len(code) == s["decode_ms_working_len"]
    assert L.oracle_ms_working_u8_len(code) == s["decode_ms_working_u8_len"]
    assert L.oracle_output_len(code) == s["output_len"]


@pytest.mark.parametrize("code", CODES, ids=oracle.CODES)
def test_encode_parity_kat(code, kats):
    """src/encoder.rs:361-527: parity bytes for data 0,1,2,... (the H-derived encoder)."""
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    assert cw[oracle.k(code) // 8:].tolist() == kats["encode_parity"][oracle.CODES[code]]


def test_doctest_encode(kats):
    """src/lib.rs:135-143"""
    d = kats["doctest_encode"]
    cw = oracle.copy_encode(0, np.arange(8, dtype=np.uint8))
    assert cw.tolist() == d["codeword"]


def test_hard_llr_vectors(kats):
    """src/decoder.rs:553-605"""
    h = kats["hard_llr"]
    hard = np.array(h["hard"], dtype=np.uint8)
    for dt in (np.int8, np.int16, np.int32, np.float32, np.float64):
        llrs = oracle.hard_to_llrs(0, hard, dt)
        assert llrs.tolist() == [dt(x) for x in h["llrs"]]
        assert oracle.llrs_to_hard(0, llrs).tolist() == h["hard"]


@pytest.mark.parametrize("code", CODES, ids=oracle.CODES)
@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.float32, np.float64])
def test_decode_ms_three_flips(code, dtype, kats):
    """src/decoder.rs:671-699 (i8) and benches/decode.rs:39-71 (f32): corrects 3 flipped bits in
    at most 50 iterations.  Iteration counts 2 (TC) / 3 (TM) are this build's own goldens."""
    sc = kats["decode_scenario"]
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    rx = cw.copy()
    rx[0] ^= sc["flip_byte0_mask"]
    ok, iters, out = oracle.decode_ms(code, oracle.hard_to_llrs(code, rx, dtype), sc["maxiters"])
    assert ok and (out[: oracle.n(code) // 8] == cw).all()
    assert iters == (2 if code < 3 else 3)
    assert oracle.syndrome_weight(code, out) == 0


@pytest.mark.parametrize("code", [c for c in CODES if c >= 3], ids=oracle.CODES[3:])
def test_decode_ms_clean_punctured(code):
    """src/decoder.rs:632-643: on the clean codeword decode_ms succeeds and its output including
    the punctured bits is a codeword of H (what the erasure decoder reconstructs)."""
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    ok, iters, out = oracle.decode_ms(code, oracle.hard_to_llrs(code, cw, np.int8), 50)
    assert ok and iters == 1
    assert (out[: oracle.n(code) // 8] == cw).all() and oracle.syndrome_weight(code, out) == 0


def test_maxiters_zero():
    """src/decoder.rs:380, :466-474: no iteration -> all-zero output, (false, 0)."""
    llrs = -np.ones(oracle.n(0), dtype=np.float32)
    ok, iters, out = oracle.decode_ms(0, llrs, 0)
    assert not ok and iters == 0 and not out.any()


def test_batch_driver_matches_single():
    rng = np.random.default_rng(3)
    llrs, _ = oracle.awgn_llrs(2, rng, 16, 2.0, np.float32)
    out, iters, ok, used = oracle.decode_ms_batch(2, llrs, 25, 2)
    assert used == 2
    for f in range(16):
        o1, i1, out1 = oracle.decode_ms(2, llrs[f], 25)
        assert (o1, i1) == (bool(ok[f]), int(iters[f])) and (out1 == out[f]).all()


@pytest.mark.parametrize("code", CODES, ids=oracle.CODES)
def test_decode_bf_three_flips(code, kats):
    """src/decoder.rs:647-670: the bit-flipping decoder recovers three flipped bits within 50 iterations."""
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    rx = cw.copy()
    rx[0] ^= kats["decode_scenario"]["flip_byte0_mask"]
    ok, iters, out = oracle.decode_bf(code, rx, 50)
    assert ok and (out[: oracle.n(code) // 8] == cw).all()
    assert oracle.syndrome_weight(code, out) == 0


@pytest.mark.parametrize("code", [c for c in CODES if c >= 3], ids=oracle.CODES[3:])
def test_decode_erasures_matches_min_sum(code):
    """src/decoder.rs:607-645: on a clean codeword the erasure pre-pass succeeds and reconstructs the
    punctured bits exactly as decode_ms does."""
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    full = np.zeros(oracle.output_len(code), dtype=np.uint8)
    full[: len(cw)] = cw
    ok, iters, out = oracle.decode_erasures(code, full, 50)
    ok_ms, _, out_ms = oracle.decode_ms(code, oracle.hard_to_llrs(code, cw, np.int8), 50)
    assert ok and ok_ms and (out == out_ms).all()


def test_decode_bf_doctest_round_trip():
    """src/lib.rs:21-50: encode, flip one bit, decode_bf with 20 iterations recovers the data."""
    code = 0                                              # TC128
    cw = oracle.copy_encode(code, np.arange(oracle.k(code) // 8, dtype=np.uint8))
    rx = cw.copy()
    rx[0] ^= 1 << 7
    ok, iters, out = oracle.decode_bf(code, rx, 20)
    assert ok and (out[: oracle.k(code) // 8] == cw[: oracle.k(code) // 8]).all()


def test_baseline_config1_tc128_one_codeword_50_iterations_3db():
    """BASELINE.json configs[0], verbatim: TC128 (k=128 r=1/2) decode_ms on the CPU path, ONE codeword, 50 iterations, AWGN at
    Eb/N0 = 3 dB -- the reference's own CPU-runnable shape (capi/src/lib.rs:83-95: one frame, caller-owned buffers).  The frame
    is the one bench.py's config-1 leg decodes (seed 0x1DBC + code); the C oracle and the numpy restatement agree on it, and the
    result is frozen here (the GPU's single-frame entry must reproduce it: tests/test_gpu_parity.py)."""
    import hashlib
    import sys
    sys.path.insert(0, oracle.ORACLE_DIR)
    import ms_numpy
    code = 0
    rng = np.random.default_rng(0x1DBC + code)
    llrs, cws = oracle.awgn_llrs(code, rng, 1, 3.0, np.float32)
    assert hashlib.sha256(llrs.tobytes()).hexdigest()[:16] == "6e3581acdada44cc"        # the frame itself is pinned
    ok, iters, out = oracle.decode_ms(code, llrs[0], 50)
    chk, var = oracle.edges(code)
    o2, i2, k2 = ms_numpy.decode_ms(ms_numpy.Structure(chk, var, oracle.n(code) + oracle.p(code)), llrs, oracle.n(code), 50)
    assert ok and iters == 3 and out.tobytes().hex() == "1883fafcd83eb9577b082a4694f949c4"
    assert (o2[0] == out).all() and int(i2[0]) == 3 and int(k2[0]) == 1
    assert (out[: oracle.n(code) // 8] == cws[0]).all()                                  # and it is the transmitted codeword
