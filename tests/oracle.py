"""ctypes binding of the CPU oracle (oracle/libldpc_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_ROOT, "oracle")
# LDPC_ORACLE_LIB: another build of the same sources (tests/test_sanitizers.py runs the known-answer suite on the ASan + UBSan one)
ORACLE_LIB = os.environ.get("LDPC_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libldpc_oracle.so")

CODES = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]
_SUF = {np.dtype(np.int8): "i8", np.dtype(np.int16): "i16", np.dtype(np.int32): "i32",
        np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}


def build(force: bool = False) -> str:
    if os.environ.get("LDPC_ORACLE_LIB"):
        return ORACLE_LIB                          # the caller built it
    if force or not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_LIB


def build_native() -> str:
    """The same sources built -O3 -march=native ON THIS HOST (oracle/Makefile target `native`), for
    bench.py's cpu_baseline leg only; the portable build stays the one the tests check against."""
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B", "native"])     # -B: never trust a copy built on another host
    return os.path.join(ORACLE_DIR, "libldpc_oracle_native.so")


def _load(path=None):
    if path is None:
        build()
        path = ORACLE_LIB
    L = ctypes.CDLL(path)
    for f in ("code_n", "code_k", "code_punctured_bits", "code_submatrix_size", "code_circulant_size",
              "code_paritycheck_sum", "bf_working_len", "ms_working_len", "ms_working_u8_len", "output_len"):
        fn = getattr(L, "oracle_" + f)
        fn.restype, fn.argtypes = ctypes.c_size_t, [ctypes.c_int]
    L.oracle_edge_crc.restype, L.oracle_edge_crc.argtypes = ctypes.c_uint32, [ctypes.c_int]
    L.oracle_edges.restype = ctypes.c_size_t
    L.oracle_edges.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.oracle_syndrome_weight.restype = ctypes.c_size_t
    L.oracle_syndrome_weight.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.oracle_encode.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.oracle_copy_encode.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.oracle_decode_bf.restype = ctypes.c_int
    L.oracle_decode_bf.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t)]
    L.oracle_decode_erasures.restype = ctypes.c_int
    L.oracle_decode_erasures.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t)]
    for s in _SUF.values():
        fn = getattr(L, "oracle_decode_ms_" + s)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        fn = getattr(L, "oracle_decode_ms_batch_" + s)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        for h in ("hard_to_llrs_", "llrs_to_hard_"):
            fn = getattr(L, "oracle_" + h + s)
            fn.restype, fn.argtypes = None, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


L = _load()


def n(code): return L.oracle_code_n(int(code))
def k(code): return L.oracle_code_k(int(code))
def p(code): return L.oracle_code_punctured_bits(int(code))
def output_len(code): return L.oracle_output_len(int(code))


def edges(code):
    E = L.oracle_code_paritycheck_sum(int(code))
    chk = np.empty(E, dtype=np.uint16)
    var = np.empty(E, dtype=np.uint16)
    L.oracle_edges(int(code), chk.ctypes.data, var.ctypes.data, E)
    return chk, var


def copy_encode(code, data: np.ndarray) -> np.ndarray:
    cw = np.zeros(n(code) // 8, dtype=np.uint8)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    assert data.shape == (k(code) // 8,)
    assert L.oracle_copy_encode(int(code), data.ctypes.data, cw.ctypes.data) == 0
    return cw


def decode_ms(code, llrs: np.ndarray, maxiters: int):
    """Single codeword through the reference-shaped entry point. Returns (success, iters, output)."""
    code = int(code)
    llrs = np.ascontiguousarray(llrs)
    s = _SUF[llrs.dtype]
    out = np.zeros(output_len(code), dtype=np.uint8)
    w = np.zeros(L.oracle_ms_working_len(code), dtype=llrs.dtype)
    w8 = np.zeros(L.oracle_ms_working_u8_len(code), dtype=np.uint8)
    it = ctypes.c_size_t(0)
    ok = getattr(L, "oracle_decode_ms_" + s)(code, llrs.ctypes.data, out.ctypes.data, w.ctypes.data,
                                             w8.ctypes.data, maxiters, ctypes.byref(it))
    assert ok >= 0
    return bool(ok), int(it.value), out


def decode_bf(code, hard: np.ndarray, maxiters: int):
    """(success, iters, output) of the bit-flipping decoder on n/8 hard bytes."""
    code = int(code)
    hard = np.ascontiguousarray(hard, dtype=np.uint8)
    assert hard.shape == (n(code) // 8,)
    out = np.zeros(output_len(code), dtype=np.uint8)
    w = np.zeros(L.oracle_bf_working_len(code), dtype=np.uint8)
    it = ctypes.c_size_t(0)
    ok = L.oracle_decode_bf(code, hard.ctypes.data, out.ctypes.data, w.ctypes.data, maxiters, ctypes.byref(it))
    assert ok >= 0
    return bool(ok), int(it.value), out


def decode_erasures(code, codeword_np: np.ndarray, maxiters: int):
    code = int(code)
    cw = np.ascontiguousarray(codeword_np, dtype=np.uint8).copy()
    assert cw.shape == (output_len(code),)
    w = np.zeros(L.oracle_bf_working_len(code), dtype=np.uint8)
    it = ctypes.c_size_t(0)
    ok = L.oracle_decode_erasures(code, cw.ctypes.data, w.ctypes.data, maxiters, ctypes.byref(it))
    return bool(ok), int(it.value), cw


def decode_ms_batch(code, llrs: np.ndarray, maxiters: int, nthreads: int = 0, lib=None):
    """[batch, n] -> (output[batch, output_len], iters[batch] u32, success[batch] u8, threads used).
    `lib`: an alternative build of the oracle loaded with _load(path) (bench.py's native build)."""
    code = int(code)
    llrs = np.ascontiguousarray(llrs)
    assert llrs.ndim == 2 and llrs.shape[1] == n(code)
    s = _SUF[llrs.dtype]
    B = llrs.shape[0]
    out = np.zeros((B, output_len(code)), dtype=np.uint8)
    iters = np.zeros(B, dtype=np.uint32)
    succ = np.zeros(B, dtype=np.uint8)
    used = getattr(lib or L, "oracle_decode_ms_batch_" + s)(code, llrs.ctypes.data, out.ctypes.data, iters.ctypes.data,
                                                     succ.ctypes.data, B, maxiters, nthreads)
    assert used > 0
    return out, iters, succ, used


def hard_to_llrs(code, hard: np.ndarray, dtype) -> np.ndarray:
    llrs = np.zeros(n(code), dtype=dtype)
    hard = np.ascontiguousarray(hard, dtype=np.uint8)
    getattr(L, "oracle_hard_to_llrs_" + _SUF[np.dtype(dtype)])(int(code), hard.ctypes.data, llrs.ctypes.data)
    return llrs


def llrs_to_hard(code, llrs: np.ndarray) -> np.ndarray:
    out = np.zeros(n(code) // 8, dtype=np.uint8)
    llrs = np.ascontiguousarray(llrs)
    getattr(L, "oracle_llrs_to_hard_" + _SUF[llrs.dtype])(int(code), llrs.ctypes.data, out.ctypes.data)
    return out


def syndrome_weight(code, bits_np: np.ndarray) -> int:
    bits_np = np.ascontiguousarray(bits_np, dtype=np.uint8)
    assert bits_np.shape == (output_len(code),)
    return int(L.oracle_syndrome_weight(int(code), bits_np.ctypes.data))


def awgn_llrs(code, rng: np.random.Generator, frames: int, ebn0_db: float, dtype=np.float32,
              scale: float = 8.0, lim: int = 31):
    """Seeded synthetic frames as BASELINE.md section 3 defines them. Returns (llrs, codewords)."""
    N, K = n(code), k(code)
    sigma = float(np.sqrt(1.0 / (2.0 * (K / N) * 10.0 ** (ebn0_db / 10.0))))
    cws = np.zeros((frames, N // 8), dtype=np.uint8)
    for f in range(frames):
        cws[f] = copy_encode(code, rng.integers(0, 256, K // 8, dtype=np.uint8))
    bits = np.unpackbits(cws, axis=1)
    y = (1.0 - 2.0 * bits) + sigma * rng.standard_normal((frames, N))
    if np.dtype(dtype).kind == "f":
        return y.astype(dtype), cws
    info = np.iinfo(dtype)
    return np.clip(np.rint(scale * y), max(-lim, info.min), min(lim, info.max)).astype(dtype), cws
