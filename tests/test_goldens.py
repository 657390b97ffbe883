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
