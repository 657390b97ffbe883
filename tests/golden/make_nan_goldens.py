#!/usr/bin/env python3
"""Generate tests/golden/nan_<code>_<T>.npz (T in {f32, f64}): frames that contain NaN LLRs, with the results decode_ms
must give.

Why: the reference's hard_bit for floats is `self < 0.0` (src/decoder.rs:76, :85), so a NaN is "non-negative" whatever its
sign bit, `abs()` of a NaN is a NaN that no `<` lets into min1 / min2 (:430-434), and `NaN == min1` is false (:391): an LLR
NaN makes its variable's marginal and every message along its edges NaN for the whole decode, its hard decision 0, and its
neighbours see it as an erased edge.  The GPU kernels read signs from bit 31 and minima from v_min3_f32, which agrees with
all of that only if NaNs are canonicalised on load -- round 2's review (missing #4) found no test feeding one.

Expected results come from the C oracle AND the numpy restatement; nothing is written unless they agree bit for bit.
Frames per file: AWGN at the waterfall with one NaN of each flavour (quiet / signalling, sign bit clear / set), several NaNs
over different block columns, whole-NaN frames of either sign, NaN next to infinities and near-overflow LLRs, clean
codewords with a NaN on a transmitted 0 and on a transmitted 1, high-SNR frames with a few NaNs; max_iters 25, 4, 0.

Run from the repo root:  python tests/golden/make_nan_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle          # noqa: E402
import ms_numpy        # noqa: E402
from make_awgn_goldens import WATERFALL_DB, MAXITERS   # noqa: E402

KINDS = ("qnan_pos", "qnan_neg", "snan_pos", "snan_neg", "several_mixed", "all_nan_pos", "all_nan_neg", "nan_and_inf",
         "nan_and_huge", "clean_nan_on_0", "clean_nan_on_1", "easy_1", "easy_2", "easy_3", "easy_5", "nan_every_7th")


def nan_values(dtype):
    """(quiet +, quiet -, signalling +, signalling -) as scalars of `dtype`, exact bit patterns"""
    if dtype == np.float32:
        bits = np.array([0x7FC00000, 0xFFC00000, 0x7FA00000, 0xFFA00001], dtype=np.uint32)
        return bits.view(np.float32)
    bits = np.array([0x7FF8000000000000, 0xFFF8000000000000, 0x7FF4000000000000, 0xFFF4000000000001], dtype=np.uint64)
    return bits.view(np.float64)


def put(row, positions, values):
    """store NaN bit patterns without letting numpy arithmetic quiet them"""
    u = row.view(np.uint32 if row.dtype == np.float32 else np.uint64)
    v = np.asarray(values).view(u.dtype)
    for i, pos in enumerate(positions):
        u[pos] = v[i % len(v)]


def make_frames(code, name, dtype, rng):
    n, k = oracle.n(code), oracle.k(code)
    w = WATERFALL_DB[name]
    qp, qn, sp, sn = nan_values(dtype)
    y, _ = oracle.awgn_llrs(code, rng, len(KINDS), w, np.float64)
    ye, cwe = oracle.awgn_llrs(code, rng, len(KINDS), w + 3.0, np.float64)
    cw = oracle.copy_encode(code, rng.integers(0, 256, k // 8, dtype=np.uint8))
    bits = np.unpackbits(cw)
    llrs = np.zeros((len(KINDS), n), dtype=dtype)
    for i, kind in enumerate(KINDS):
        row = y[i].astype(dtype)
        pos = rng.permutation(n)
        if kind == "qnan_pos":
            put(row, pos[:1], [qp])
        elif kind == "qnan_neg":
            put(row, pos[:1], [qn])
        elif kind == "snan_pos":
            put(row, pos[:1], [sp])
        elif kind == "snan_neg":
            put(row, pos[:1], [sn])
        elif kind == "several_mixed":
            put(row, pos[:12], [qp, qn, sp, sn])
        elif kind == "all_nan_pos":
            put(row, range(n), [qp])
        elif kind == "all_nan_neg":
            put(row, range(n), [qn, sn])
        elif kind == "nan_and_inf":
            row[pos[12:20]] = np.inf
            row[pos[20:28]] = -np.inf
            put(row, pos[:6], [qn, qp, sn])
        elif kind == "nan_and_huge":
            row = (y[i] * (1e37 if dtype == np.float32 else 1e307)).astype(dtype)
            put(row, pos[:6], [qn, sp])
        elif kind == "clean_nan_on_0":
            row = oracle.hard_to_llrs(code, cw, dtype)
            put(row, [int(np.nonzero(bits == 0)[0][5])], [qn])
        elif kind == "clean_nan_on_1":
            row = oracle.hard_to_llrs(code, cw, dtype)
            put(row, [int(np.nonzero(bits == 1)[0][5])], [qn])
        elif kind.startswith("easy_"):
            row = ye[i].astype(dtype)
            zeros = np.nonzero(np.unpackbits(cwe[i]) == 0)[0]          # NaNs on transmitted zeros: the frame can still converge
            put(row, rng.permutation(zeros)[: int(kind[5:])], [qn, sp, qp, sn])
        else:
            put(row, range(0, n, 7), [qn, qp])
        llrs[i] = row
    return llrs


def main():
    for code, name in enumerate(oracle.CODES):
        n = oracle.n(code)
        chk, var = oracle.edges(code)
        st = ms_numpy.Structure(chk, var, n + oracle.p(code))
        for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
            rng = np.random.default_rng([0x4A4E, code, 0 if tag == "f32" else 1])
            llrs = make_frames(code, name, dtype, rng)
            assert np.isnan(llrs).any(axis=1).all()
            arrays = {"llrs_bits": llrs.view(np.uint32 if dtype == np.float32 else np.uint64), "kinds": np.array(KINDS),
                      "maxiters": np.array(MAXITERS)}
            with np.errstate(all="ignore"):
                for mi in MAXITERS:
                    a = oracle.decode_ms_batch(code, llrs, mi)[:3]
                    b = ms_numpy.decode_ms(st, llrs, n, mi)
                    for x, y_, what in zip(a, b, ("output", "iters", "success")):
                        if not (np.asarray(x) == np.asarray(y_)).all():
                            bad = np.nonzero((np.asarray(x) != np.asarray(y_)).reshape(len(llrs), -1).any(axis=1))[0]
                            raise SystemExit(f"{name} {tag} maxiters {mi}: the two restatements disagree on {what}, frames {bad} "
                                             f"({[KINDS[j] for j in bad]})")
                    arrays[f"output_{mi}"], arrays[f"iters_{mi}"], arrays[f"success_{mi}"] = a[0], a[1].astype(np.uint32), a[2]
            np.savez_compressed(os.path.join(HERE, f"nan_{name}_{tag}.npz"), **arrays)
            ok, it = arrays["success_25"], arrays["iters_25"]
            print(name, tag, "converged", int(ok.sum()), "of", len(ok), "iters", it.tolist(), flush=True)


if __name__ == "__main__":
    main()
