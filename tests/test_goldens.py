"""Frozen parity data: tests/golden/awgn_<code>_<T>.npz hold seeded noisy frames (converging early / late,
failing, saturating, denormal, signed-zero, all-zero, clean, three-flip) with the (output, iters, success)
decode_ms must return for max_iters 25, 4 and 0 -- written by tests/golden/make_awgn_goldens.py only when
two independently written CPU restatements of src/decoder.rs:347-475 agreed on every frame.

CPU: both restatements still reproduce the files (an edit to either shows up here).
GPU: the HIP kernels reproduce the files through the C ABI -- no oracle in the loop."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "awgn_*.npz")))
IDS = [os.path.basename(f)[5:-4] for f in FILES]
CODES = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]
# frames with NaN LLRs (quiet / signalling, either sign bit, whole-NaN frames, NaN next to inf): make_nan_goldens.py
NAN_FILES = sorted(glob.glob(os.path.join(HERE, "golden", "nan_*.npz")))
NAN_IDS = [os.path.basename(f)[4:-4] for f in NAN_FILES]


def load(path):
    z = np.load(path)
    name, tag = os.path.basename(path)[5:-4].split("_")
    return CODES.index(name), tag, z


def test_fixture_set_is_complete():
    assert len(FILES) == 27, "9 codes x {f32, i8, i32}"
    for f in FILES:
        code, tag, z = load(f)
        assert z["llrs"].dtype == {"f32": np.float32, "i8": np.int8, "i32": np.int32}[tag]
        assert z["maxiters"].tolist() == [25, 4, 0]
        ok = z["success_25"]
        assert 0 < ok.sum() < len(ok), "every file holds converging and failing frames"
        assert len(set(z["iters_25"][ok == 1].tolist())) >= 4, "a spread of iteration counts"
        assert (z["iters_0"] == 0).all() and (z["success_0"] == 0).all() and (z["output_0"] == 0).all()


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_c_oracle_reproduces_goldens(path):
    import oracle
    code, tag, z = load(path)
    for mi in z["maxiters"].tolist():
        out, it, ok, _ = oracle.decode_ms_batch(code, z["llrs"], mi)
        assert (out == z[f"output_{mi}"]).all() and (it == z[f"iters_{mi}"]).all() and (ok == z[f"success_{mi}"]).all()


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_numpy_restatement_reproduces_goldens(path):
    import sys
    import oracle
    sys.path.insert(0, oracle.ORACLE_DIR)
    import ms_numpy
    code, tag, z = load(path)
    chk, var = oracle.edges(code)
    st = ms_numpy.Structure(chk, var, oracle.n(code) + oracle.p(code))
    with np.errstate(over="ignore"):
        for mi in (25, 4):
            out, it, ok = ms_numpy.decode_ms(st, z["llrs"], oracle.n(code), mi)
            assert (out == z[f"output_{mi}"]).all() and (it == z[f"iters_{mi}"]).all() and (ok == z[f"success_{mi}"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_hip_reproduces_goldens(path):
    import torch
    from labrador_ldpc_amd import LDPCCode
    code_id, tag, z = load(path)
    code = LDPCCode(code_id)
    llrs = z["llrs"]
    d = torch.from_numpy(llrs).cuda()
    for mi in z["maxiters"].tolist():
        want = (z[f"output_{mi}"], z[f"iters_{mi}"], z[f"success_{mi}"])
        got_h = code.decode_ms_batch(llrs, mi)                                     # host pointers, staged
        got_d = [t.cpu().numpy() for t in code.decode_ms_batch(d, mi)]             # device-resident
        for g_h, g_d, w, what in zip(got_h, got_d, want, ("output", "iters", "success")):
            assert (g_h == w).all(), f"{code.name} {tag} max_iters {mi}: host-path {what} differs from the golden file"
            assert (g_d.astype(w.dtype) == w).all(), f"{code.name} {tag} max_iters {mi}: device-path {what} differs"
    # frame by frame through the reference-shaped single-codeword entry point (capi/src/lib.rs:83-127)
    for f in range(0, len(llrs), 7):
        out = np.zeros(code.output_len(), dtype=np.uint8)
        ok, it = code.decode_ms(llrs[f], out, maxiters=25)
        assert ok == bool(z["success_25"][f]) and it == int(z["iters_25"][f]) and (out == z["output_25"][f]).all()


# ---- NaN LLRs: hard_bit is `x < 0.0` (src/decoder.rs:76, :85), false for a NaN whatever its sign bit --------------------
def load_nan(path):
    z = np.load(path)
    name, tag = os.path.basename(path)[4:-4].split("_")
    llrs = z["llrs_bits"].view(np.float32 if tag == "f32" else np.float64)      # stored as integers: the exact NaN patterns
    return CODES.index(name), tag, z, llrs


def test_nan_fixture_set_is_complete():
    assert len(NAN_FILES) == 18, "9 codes x {f32, f64}"
    for f in NAN_FILES:
        code, tag, z, llrs = load_nan(f)
        assert np.isnan(llrs).any(axis=1).all()
        bits = z["llrs_bits"]
        top = bits >> (bits.dtype.itemsize * 8 - 1)
        assert (np.isnan(llrs) & (top == 1)).any() and (np.isnan(llrs) & (top == 0)).any(), "NaNs of both signs"
        ok = z["success_25"]
        assert 0 < ok.sum() < len(ok)
        kinds = z["kinds"].tolist()
        # a whole frame of NaNs decodes to the all-zero codeword at iteration 0 (every hard bit 0, every parity even)
        for k in ("all_nan_pos", "all_nan_neg"):
            i = kinds.index(k)
            assert ok[i] == 1 and z["iters_25"][i] == 0 and not z["output_25"][i].any()


@pytest.mark.parametrize("path", NAN_FILES, ids=NAN_IDS)
def test_both_restatements_reproduce_nan_goldens(path):
    import sys
    import oracle
    sys.path.insert(0, oracle.ORACLE_DIR)
    import ms_numpy
    code, tag, z, llrs = load_nan(path)
    chk, var = oracle.edges(code)
    st = ms_numpy.Structure(chk, var, oracle.n(code) + oracle.p(code))
    with np.errstate(all="ignore"):
        for mi in z["maxiters"].tolist():
            out, it, ok, _ = oracle.decode_ms_batch(code, llrs, mi)
            assert (out == z[f"output_{mi}"]).all() and (it == z[f"iters_{mi}"]).all() and (ok == z[f"success_{mi}"]).all()
            if mi:
                out, it, ok = ms_numpy.decode_ms(st, llrs, oracle.n(code), mi)
                assert (out == z[f"output_{mi}"]).all() and (it == z[f"iters_{mi}"]).all() and (ok == z[f"success_{mi}"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", NAN_FILES, ids=NAN_IDS)
def test_hip_reproduces_nan_goldens(path):
    import torch
    from labrador_ldpc_amd import LDPCCode
    code_id, tag, z, llrs = load_nan(path)
    code = LDPCCode(code_id)
    d = torch.from_numpy(z["llrs_bits"].astype(np.int32 if tag == "f32" else np.int64)).cuda().view(torch.float32 if tag == "f32" else torch.float64)
    variants = [0, 256]                                   # queue-fed and fixed-stride distribution
    if tag == "f32":
        variants += {"TM8192": [2, 4], "TM2048": [2, 32], "TM1536": [2], "TM6144": [2]}.get(code.name, [])
    else:
        variants += [100]                                 # the f64 workspace kernel
    for mi in z["maxiters"].tolist():
        want = (z[f"output_{mi}"], z[f"iters_{mi}"], z[f"success_{mi}"])
        for variant in variants:
            got_d = [t.cpu().numpy() for t in code.decode_ms_batch(d, mi, variant=variant)]
            for g_d, w, what in zip(got_d, want, ("output", "iters", "success")):
                bad = np.nonzero((g_d.astype(w.dtype) != w).reshape(len(w), -1).any(axis=1))[0]
                assert bad.size == 0, (f"{code.name} {tag} max_iters {mi} variant {variant}: {what} differs on frames "
                                       f"{bad.tolist()} ({[z['kinds'][i] for i in bad]})")
        got_h = code.decode_ms_batch(llrs, mi)            # host pointers, staged
        for g_h, w in zip(got_h, want):
            assert (g_h == w).all()
    for f in range(len(llrs)):                            # the reference-shaped single-codeword entry point
        out = np.zeros(code.output_len(), dtype=np.uint8)
        ok, it = code.decode_ms(llrs[f], out, maxiters=25)
        assert ok == bool(z["success_25"][f]) and it == int(z["iters_25"][f]) and (out == z["output_25"][f]).all()
