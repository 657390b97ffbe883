"""CPU-side checks of the product library: it loads, exports every symbol the header declares,
its code tables reproduce the reference's known answers, host helpers behave like the crate's,
and the decoders fail loudly (no CPU fallback) when there is no GPU.  No compute calls need a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import labrador_ldpc_amd as la
from labrador_ldpc_amd import LDPCCode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "labrador_ldpc_hip.h")).read()
    declared = set(re.findall(r"\b(labrador_ldpc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    dll = ctypes.CDLL(la.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(dll, name), f"{name} declared in the header but not exported"
    assert declared == set(la.SYMBOLS), "python binding and header disagree"


@pytest.mark.parametrize("code", list(LDPCCode), ids=NAMES)
def test_tables_reproduce_reference_kats(code, kats):
    """The constexpr tables the kernels are generated from: edge-order CRC (src/codes/mod.rs:521-523),
    sizes (src/codes/mod.rs:109-241), encoder parity (src/encoder.rs:361-527)."""
    s = kats["sizes"][code.name]
    assert la.lib.labrador_ldpc_hip_edge_crc(int(code)) == kats["edge_crc"][int(code)]
    assert (code.n(), code.k(), code.punctured_bits()) == (s["n"], s["k"], s["punctured_bits"])
    assert code.submatrix_size() == s["submatrix_size"] and code.circulant_size() == s["circulant_size"]
    assert code.paritycheck_sum() == s["paritycheck_sum"]
    assert code.decode_bf_working_len() == s["decode_bf_working_len"]
    assert code.decode_ms_working_len() == s["decode_ms_working_len"]
    assert code.decode_ms_working_u8_len() == s["decode_ms_working_u8_len"]
    assert code.output_len() == s["output_len"]
    data = np.arange(code.k() // 8, dtype=np.uint8)
    cw = np.zeros(code.n() // 8, dtype=np.uint8)
    code.copy_encode(data, cw)
    assert cw[code.k() // 8:].tolist() == kats["encode_parity"][code.name]
    cw2 = np.zeros(code.n() // 8, dtype=np.uint8)
    cw2[: code.k() // 8] = data
    code.encode(cw2)
    assert (cw2 == cw).all()


@pytest.mark.parametrize("code", list(LDPCCode), ids=NAMES)
def test_iter_paritychecks_list(code, kats):
    """The product's own edge list: count == paritycheck_sum (src/codes/mod.rs:532), CRC of the listed pairs ==
    the reference's known answer (:521-523), and identical to the oracle's independently written enumerator."""
    import oracle
    chk, var = code.iter_paritychecks()
    assert len(chk) == code.paritycheck_sum() == kats["sizes"][code.name]["paritycheck_sum"]
    crc = 0xFFFFFFFF
    for c, v in zip(chk.tolist(), var.tolist()):
        for word in (c, v):
            crc ^= word
            for _ in range(16):
                crc = (crc >> 1) ^ (0xEDB88320 if crc & 1 else 0)
    assert crc == kats["edge_crc"][int(code)]
    o_chk, o_var = oracle.edges(code)
    assert (chk == o_chk).all() and (var == o_var).all()
    assert int(var.max()) == code.n() + code.punctured_bits() - 1 and int(chk.max()) == code.n() + code.punctured_bits() - code.k() - 1
    assert la.lib.labrador_ldpc_hip_edges(int(code), None, None, 0) == len(chk)        # count-only call
    assert la.lib.labrador_ldpc_hip_edges(99, None, None, 0) == 0


def test_encoder_matches_oracle_on_random_data():
    import oracle
    rng = np.random.default_rng(1)
    for code in LDPCCode:
        data = rng.integers(0, 256, code.k() // 8, dtype=np.uint8)
        cw = np.zeros(code.n() // 8, dtype=np.uint8)
        code.copy_encode(data, cw)
        assert (cw == oracle.copy_encode(code, data)).all()


def test_hard_llr_helpers(kats):
    h = kats["hard_llr"]
    code = LDPCCode.TC128
    hard = np.array(h["hard"], dtype=np.uint8)
    for dt in (np.int8, np.int16, np.float32, np.float64):
        llrs = np.zeros(code.n(), dtype=dt)
        code.hard_to_llrs(hard, llrs)
        assert llrs.tolist() == [dt(x) for x in h["llrs"]]
        back = np.zeros(code.n() // 8, dtype=np.uint8)
        code.llrs_to_hard(llrs, back)
        assert back.tolist() == h["hard"]


def test_out_of_range_code_is_rejected():
    for bad in (-1, 9, 1000):
        assert la.lib.labrador_ldpc_code_n(bad) == 0 and la.lib.labrador_ldpc_output_len(bad) == 0
        assert la.lib.labrador_ldpc_hip_edge_crc(bad) == 0
    out = np.zeros(4, dtype=np.uint8)
    st = la.lib.labrador_ldpc_decode_ms_batch_f32(9, out.ctypes.data, out.ctypes.data, out.ctypes.data,
                                                  out.ctypes.data, 1, 10, None)
    assert st == -1 and "out of range" in la.last_error()


def test_length_checks_mirror_the_crate_asserts():
    code = LDPCCode.TC128
    with pytest.raises(ValueError):
        code.decode_ms(np.zeros(code.n() - 1, dtype=np.float32), np.zeros(code.output_len(), dtype=np.uint8))
    with pytest.raises(ValueError):
        code.decode_ms(np.zeros(code.n(), dtype=np.float32), np.zeros(code.output_len() + 1, dtype=np.uint8))
    with pytest.raises(ValueError):
        code.decode_ms(np.zeros(code.n(), dtype=np.float32), np.zeros(code.output_len(), dtype=np.uint8),
                       working=np.zeros(3, dtype=np.float32))
    with pytest.raises(ValueError):
        code.encode(np.zeros(code.n() // 8 + 1, dtype=np.uint8))
    for bad in ([0.0] * code.n(), np.zeros(code.n(), dtype=np.uint16), "llrs"):
        with pytest.raises(ValueError):
            code.decode_ms(bad, np.zeros(code.output_len(), dtype=np.uint8))


def test_decode_without_gpu_fails_loudly():
    """No CPU fallback: without a gfx950 device the batched call returns ENODEV and the
    reference-shaped call raises."""
    if la.device_count() > 0:
        pytest.skip("a GPU is present")
    code = LDPCCode.TC128
    llrs = np.ones((2, code.n()), dtype=np.float32)
    with pytest.raises(la.LdpcHipError) as e:
        code.decode_ms_batch(llrs, 10)
    assert "status -2" in str(e.value)
    with pytest.raises(la.LdpcHipError):
        code.decode_ms(llrs[0], np.zeros(code.output_len(), dtype=np.uint8))
    with pytest.raises(la.LdpcHipError):
        code.decode_bf(np.zeros(code.n() // 8, dtype=np.uint8), np.zeros(code.output_len(), dtype=np.uint8))
    with pytest.raises(la.LdpcHipError):
        code.encode_batch(np.zeros((2, code.k() // 8), dtype=np.uint8))


def test_abi_version_and_growable_opts_struct():
    """struct labrador_ldpc_hip_opts grew between 0.1.0 and 0.2.0 with nothing to tell the layouts apart.  ABI 3: the
    struct starts with struct_size; the library reads a field only if it lies inside the first struct_size bytes and takes the
    rest as zero, so a caller that knows fewer fields cannot hand it padding as n_devices / devices."""
    import ctypes
    assert la.lib.labrador_ldpc_hip_abi_version() == 3
    assert "0.6.0" in la.lib.labrador_ldpc_hip_version().decode()
    assert ctypes.sizeof(la.HipOpts) == 40 and la.HipOpts.device.offset == 8 and la.HipOpts.n_devices.offset == 28
    code = LDPCCode.TC128
    llrs = np.ones((2, code.n()), dtype=np.float32)
    out = np.zeros((2, code.output_len()), dtype=np.uint8)
    it, ok = np.zeros(2, dtype=np.uint32), np.zeros(2, dtype=np.uint8)
    call = lambda o: la.lib.labrador_ldpc_decode_ms_batch_f32(int(code), llrs.ctypes.data, out.ctypes.data, it.ctypes.data,
                                                              ok.ctypes.data, 2, 5, ctypes.byref(o))
    gpu = la.device_count() > 0
    # a full-size struct with a device list is checked as one (without a GPU the first check that can fail is the stream's) ...
    bad_stream = None if gpu else 1234
    want = "devices is NULL" if gpu else "opts->stream must be NULL"
    st = call(la.HipOpts(-1, la.MEM_HOST, bad_stream, 0, 5, None))
    assert st == -1 and want in la.last_error(), la.last_error()
    # ... the same bytes from a caller whose struct ends before n_devices are a plain single-device call: the 5 is not read
    st = call(la.HipOpts(-1, la.MEM_HOST, bad_stream, 0, 5, None, struct_size=la.HipOpts.n_devices.offset))
    if gpu:
        assert st == 0
    else:
        assert st == -2 and "no HIP device" in la.last_error()
    # struct_size 0 (what `= {0}` leaves) means this header's layout; a larger one (a newer client) is accepted too
    for size in (0, 64):
        st = call(la.HipOpts(-1, la.MEM_HOST, bad_stream, 0, 5, None, struct_size=size))
        assert st == -1 and want in la.last_error()


def test_single_frame_calls_define_their_outputs_when_the_library_cannot_run():
    """the reference-shaped calls can only return `false`; when the library could not run they must not leave
    *iters_run and output as they found them (the reference always writes them, capi/src/lib.rs:91-93)."""
    import ctypes
    if la.device_count() > 0:
        pytest.skip("a GPU is present: the calls run")
    code = LDPCCode.TC128
    llrs = np.ones(code.n(), dtype=np.float32)
    out = np.full(code.output_len(), 0xAB, dtype=np.uint8)
    iters = ctypes.c_size_t(12345)
    ok = la.lib.labrador_ldpc_decode_ms_f32(int(code), llrs.ctypes.data, out.ctypes.data, None, None, 50, ctypes.byref(iters))
    assert not ok and iters.value == 50 and not out.any() and "no HIP device" in la.last_error()
    out[:] = 0xCD
    iters.value = 777
    ok = la.lib.labrador_ldpc_decode_bf(int(code), np.zeros(code.n() // 8, dtype=np.uint8).ctypes.data, out.ctypes.data, None, 20,
                                        ctypes.byref(iters))
    assert not ok and iters.value == 20 and not out.any()


def test_shader_clock_probe_argument_checks_and_no_device_path():
    """labrador_ldpc_hip_shader_clock_mhz (round 6: the per-rank clock of a bench line): NULL result pointer -> EINVAL; without a GPU
    -> ENODEV with the reason in last_error(), nothing written; with one: a plausible gfx950 shader clock."""
    import ctypes
    assert la.lib.labrador_ldpc_hip_shader_clock_mhz(0, 1.0, None) == -1 and "NULL" in la.last_error()
    mhz = ctypes.c_double(-1.0)
    st = la.lib.labrador_ldpc_hip_shader_clock_mhz(0, 1.0, ctypes.byref(mhz))
    if la.device_count() == 0:
        assert st == -2 and mhz.value == -1.0 and "no HIP device" in la.last_error()
    else:
        assert st == 0 and 1000.0 < mhz.value < 3000.0
