#!/usr/bin/env python3
"""Generate tests/golden/awgn_<code>_<T>.npz: seeded noisy frames with the results decode_ms must give.

Why: no reference test asserts iteration counts, failed frames or soft-input behaviour
(src/decoder.rs:671-699 checks success + codeword only), which is everything the 2 dB
benchmark exercises.  These files freeze them as DATA: a later edit that changed the oracle and the
kernels together would no longer go unnoticed.

How the expected results are produced: by the C oracle (oracle/ldpc_decode_tmpl.h) AND by an
independently structured numpy restatement written from the reference text (oracle/ms_numpy.py); this
script refuses to write a file unless the two agree bit for bit on every frame, and checks the edge lists
it feeds the second restatement against the reference's own CRC-32 known answers
(tests/golden/reference_kats.json <- src/codes/mod.rs:517-535).

Frames per file (LLR dtype T in {f32, i8}; i32 gets a small file too): AWGN at the code's waterfall
Eb/N0, 1.5 dB above and 3 dB below (converge late / early / fail), plus special frames: all-zero LLRs, a clean
codeword, the reference's three-flip scenario (src/decoder.rs:676-688), saturating / huge / denormal /
signed-zero inputs.  Results are stored for max_iters 25 (the benchmark's), 4 and 0.

Run from the repo root:  python tests/golden/make_awgn_goldens.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle          # noqa: E402
import ms_numpy        # noqa: E402

WATERFALL_DB = {"TC128": 4.0, "TC256": 3.5, "TC512": 3.0, "TM1280": 3.6, "TM1536": 2.6, "TM2048": 2.0,
                "TM5120": 3.3, "TM6144": 2.3, "TM8192": 1.9}
MAXITERS = (25, 4, 0)


def awgn(code, rng, frames, ebn0):
    y, cws = oracle.awgn_llrs(code, rng, frames, ebn0, np.float64)
    return y, cws


def make_frames(code, name, dtype, total, rng):
    n, k = oracle.n(code), oracle.k(code)
    w = WATERFALL_DB[name]
    n_mid, n_hi, n_lo = int(total * 0.4), int(total * 0.2), int(total * 0.2)
    n_special = total - n_mid - n_hi - n_lo
    ys, kinds = [], []
    for cnt, eb, kind in ((n_mid, w, "waterfall"), (n_hi, w + 1.5, "easy"), (n_lo, w - 3.0, "hard")):
        y, _ = awgn(code, rng, cnt, eb)
        ys.append(y)
        kinds += [kind] * cnt
    y = np.concatenate(ys)
    if dtype == np.float32:
        llrs = y.astype(np.float32)
    elif dtype == np.int8:
        llrs = np.clip(np.rint(8.0 * y), -32, 31).astype(np.int8)
    else:                                                                    # i32: values up to the type's limits
        llrs = np.clip(np.rint(y * 3e8), -2**31, 2**31 - 1).astype(np.int32)
    # special frames
    sp = np.zeros((n_special, n), dtype=dtype)
    sk = []
    cw = oracle.copy_encode(code, rng.integers(0, 256, k // 8, dtype=np.uint8))
    clean = oracle.hard_to_llrs(code, cw, dtype)
    flipped = cw.copy()
    flipped[0] ^= 1 << 7 | 1 << 5 | 1 << 3                                    # decoder.rs:676-679
    yn, _ = awgn(code, rng, n_special, w)
    for i in range(n_special):
        kind = ("zeros", "clean", "three_flips", "saturating", "extreme", "tiny_or_signed_zero")[i % 6]
        if kind == "zeros":
            pass
        elif kind == "clean":
            sp[i] = clean
        elif kind == "three_flips":
            sp[i] = oracle.hard_to_llrs(code, flipped, dtype)
        elif kind == "saturating":
            if dtype == np.float32:
                sp[i] = (yn[i] * 1e30).astype(np.float32)                     # sums stay finite (< 3.4e38)
            elif dtype == np.int8:
                sp[i] = np.clip(np.rint(8.0 * yn[i]), -127, 127).astype(np.int8)   # accumulations saturate
            else:
                sp[i] = np.clip(np.rint(yn[i] * 2e9), -2**31, 2**31 - 1).astype(np.int32)
        elif kind == "extreme":
            if dtype == np.float32:
                sp[i] = (yn[i] * 3e37).astype(np.float32)                     # near FLT_MAX: some sums overflow to inf
                sp[i][~np.isfinite(sp[i])] = np.float32(3e38)
            elif dtype == np.int8:
                sp[i] = np.clip(np.rint(60.0 * yn[i]), -128, 127).astype(np.int8)  # includes -128: |-128| = 127
            else:
                sp[i] = np.where(yn[i] < 0, -2**31, 2**31 - 1).astype(np.int32)
        else:
            if dtype == np.float32:
                sp[i] = (yn[i] * 1e-41).astype(np.float32)                    # denormals
                sp[i][::7] = np.float32(-0.0)
                sp[i][3::11] = np.float32(0.0)
            else:
                sp[i] = np.sign(np.rint(yn[i] * 0.8)).astype(dtype)           # many exact zeros, +-1 elsewhere
        sk.append(kind)
    return np.concatenate([llrs, sp]), kinds + sk


def main():
    with open(os.path.join(HERE, "reference_kats.json")) as f:
        kats = json.load(f)
    summary = {}
    for code, name in enumerate(oracle.CODES):
        n = oracle.n(code)
        chk, var = oracle.edges(code)
        crc = 0xFFFFFFFF                                                      # src/codes/mod.rs:507-531
        for c, v in zip(chk.tolist(), var.tolist()):
            for word in (c, v):
                crc ^= word
                for _ in range(16):
                    crc = (crc >> 1) ^ (0xEDB88320 if crc & 1 else 0)
        assert crc == kats["edge_crc"][code], f"{name}: edge list does not reproduce the reference CRC"
        st = ms_numpy.Structure(chk, var, n + oracle.p(code))
        for dtype, tag, total in ((np.float32, "f32", 64 if n <= 2048 else 32), (np.int8, "i8", 64 if n <= 2048 else 32),
                                  (np.int32, "i32", 16 if n <= 2048 else 12)):
            rng = np.random.default_rng([0x1DBC, code, {"f32": 0, "i8": 1, "i32": 2}[tag]])
            llrs, kinds = make_frames(code, name, dtype, total, rng)
            arrays = {"llrs": llrs, "kinds": np.array(kinds), "maxiters": np.array(MAXITERS)}
            for mi in MAXITERS:
                a = oracle.decode_ms_batch(code, llrs, mi)[:3]
                b = ms_numpy.decode_ms(st, llrs, n, mi)
                for x, y, what in zip(a, b, ("output", "iters", "success")):
                    if not (np.asarray(x) == np.asarray(y)).all():
                        raise SystemExit(f"{name} {tag} maxiters {mi}: the two restatements disagree on {what}")
                arrays[f"output_{mi}"], arrays[f"iters_{mi}"], arrays[f"success_{mi}"] = a[0], a[1].astype(np.uint32), a[2]
            np.savez_compressed(os.path.join(HERE, f"awgn_{name}_{tag}.npz"), **arrays)
            it, ok = arrays["iters_25"], arrays["success_25"]
            summary[f"{name}_{tag}"] = {"frames": int(len(llrs)), "converged": int(ok.sum()),
                                        "iters_min": int(it[ok == 1].min()) if ok.any() else None,
                                        "iters_max": int(it[ok == 1].max()) if ok.any() else None,
                                        "failed": int((ok == 0).sum())}
            print(name, tag, summary[f"{name}_{tag}"], flush=True)
    with open(os.path.join(HERE, "awgn_goldens_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
