"""labrador_ldpc_amd -- MI355X (gfx950) batched min-sum LDPC decoder behind labrador-ldpc's API.

Host-side mirror of the reference crate's `LDPCCode` surface for the decode_ms path
(reference: src/codes/mod.rs:365-441 accessors, src/decoder.rs:88-116 sizes, :347-475
decode_ms, :484-509 LLR helpers, src/encoder.rs:293-315 encode/copy_encode), implemented as a
thin ctypes binding over the C ABI of ``liblabrador_ldpc_hip.so`` (include/labrador_ldpc_hip.h).

The library is required: importing this package without the built ``.so`` raises, and the
decoders raise ``LdpcHipError`` when no gfx950 device is usable.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import enum
import os
from typing import Optional, Tuple

import numpy as np

__all__ = ["LDPCCode", "LdpcHipError", "lib", "device_count", "last_error", "HipOpts", "LIB_PATH"]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblabrador_ldpc_hip.so")


class LdpcHipError(RuntimeError):
    """A call into liblabrador_ldpc_hip.so failed (status < 0)."""


class HipOpts(ctypes.Structure):
    """struct labrador_ldpc_hip_opts (include/labrador_ldpc_hip.h)."""

    _fields_ = [("struct_size", ctypes.c_size_t), ("device", ctypes.c_int), ("memory", ctypes.c_int),
                ("stream", ctypes.c_void_p), ("variant", ctypes.c_int),
                ("n_devices", ctypes.c_int), ("devices", ctypes.POINTER(ctypes.c_int))]

    def __init__(self, device=0, memory=0, stream=None, variant=0, n_devices=0, devices=None, struct_size=None):
        # struct_size (ABI 3) tells the library how much of the struct this caller knows; fields beyond it read as zero
        super().__init__(ctypes.sizeof(HipOpts) if struct_size is None else struct_size, device, memory, stream, variant,
                         n_devices, devices)


MEM_HOST, MEM_DEVICE = 0, 1
DEVICE_CURRENT, DEVICE_ALL = -1, -2

_c = ctypes
_sz, _vp, _int = _c.c_size_t, _c.c_void_p, _c.c_int
_optp = _c.POINTER(HipOpts)

# every symbol include/labrador_ldpc_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "labrador_ldpc_code_n": (_sz, [_int]),
    "labrador_ldpc_code_k": (_sz, [_int]),
    "labrador_ldpc_bf_working_len": (_sz, [_int]),
    "labrador_ldpc_ms_working_u8_len": (_sz, [_int]),
    "labrador_ldpc_ms_working_len": (_sz, [_int]),
    "labrador_ldpc_output_len": (_sz, [_int]),
    "labrador_ldpc_encode": (None, [_int, _vp]),
    "labrador_ldpc_copy_encode": (None, [_int, _vp, _vp]),
    "labrador_ldpc_decode_bf": (_c.c_bool, [_int, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_decode_ms_i8": (_c.c_bool, [_int, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_decode_ms_i16": (_c.c_bool, [_int, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_decode_ms_i32": (_c.c_bool, [_int, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_decode_ms_f32": (_c.c_bool, [_int, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_decode_ms_f64": (_c.c_bool, [_int, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)]),
    "labrador_ldpc_hard_to_llrs_i8": (None, [_int, _vp, _vp]),
    "labrador_ldpc_hard_to_llrs_i16": (None, [_int, _vp, _vp]),
    "labrador_ldpc_hard_to_llrs_i32": (None, [_int, _vp, _vp]),
    "labrador_ldpc_hard_to_llrs_f32": (None, [_int, _vp, _vp]),
    "labrador_ldpc_hard_to_llrs_f64": (None, [_int, _vp, _vp]),
    "labrador_ldpc_llrs_to_hard_i8": (None, [_int, _vp, _vp]),
    "labrador_ldpc_llrs_to_hard_i16": (None, [_int, _vp, _vp]),
    "labrador_ldpc_llrs_to_hard_i32": (None, [_int, _vp, _vp]),
    "labrador_ldpc_llrs_to_hard_f32": (None, [_int, _vp, _vp]),
    "labrador_ldpc_llrs_to_hard_f64": (None, [_int, _vp, _vp]),
    "labrador_ldpc_decode_ms_batch_f32": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    "labrador_ldpc_decode_ms_batch_i8": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    "labrador_ldpc_decode_ms_batch_i16": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    "labrador_ldpc_decode_ms_batch_i32": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    "labrador_ldpc_decode_ms_batch_f64": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    **{f"labrador_ldpc_decode_ms_batch_{t}_multi": (_int, [_int, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int]) for t in ("i8", "i16", "i32", "f32", "f64")},
    "labrador_ldpc_decode_bf_batch": (_int, [_int, _vp, _vp, _vp, _vp, _sz, _sz, _optp]),
    "labrador_ldpc_encode_batch": (_int, [_int, _vp, _vp, _sz, _optp]),
    **{f"labrador_ldpc_hard_to_llrs_batch_{t}": (_int, [_int, _vp, _vp, _sz, _optp]) for t in ("i8", "i16", "i32", "f32", "f64")},
    **{f"labrador_ldpc_llrs_to_hard_batch_{t}": (_int, [_int, _vp, _vp, _sz, _optp]) for t in ("i8", "i16", "i32", "f32", "f64")},
    "labrador_ldpc_hip_awgn_f32": (_int, [_int, _vp, _sz, _vp, _sz, _c.c_float, _c.c_uint64, _optp]),
    "labrador_ldpc_hip_awgn_i8": (_int, [_int, _vp, _sz, _vp, _sz, _c.c_float, _c.c_float, _int,
                                         _c.c_uint64, _optp]),
    "labrador_ldpc_hip_awgn_f32_at": (_int, [_int, _vp, _sz, _vp, _c.c_uint64, _sz, _c.c_float, _c.c_uint64, _optp]),
    "labrador_ldpc_hip_awgn_i8_at": (_int, [_int, _vp, _sz, _vp, _c.c_uint64, _sz, _c.c_float, _c.c_float, _int,
                                            _c.c_uint64, _optp]),
    "labrador_ldpc_hip_edge_crc": (_c.c_uint32, [_int]),
    "labrador_ldpc_hip_edges": (_sz, [_int, _vp, _vp, _sz]),
    "labrador_ldpc_hip_shard_range": (_int, [_sz, _sz, _sz, _c.POINTER(_sz), _c.POINTER(_sz)]),
    "labrador_ldpc_hip_device_count": (_int, []),
    "labrador_ldpc_hip_last_error": (_c.c_char_p, []),
    "labrador_ldpc_hip_version": (_c.c_char_p, []),
    "labrador_ldpc_hip_build_id": (_c.c_char_p, []),
    "labrador_ldpc_hip_abi_version": (_int, []),
    "labrador_ldpc_hip_shader_clock_mhz": (_int, [_int, _c.c_double, _c.POINTER(_c.c_double)]),
    "labrador_ldpc_hip_decode_ms_i8_kernel": (_c.c_char_p, [_int, _int, _sz]),
}


def _preload_hip_runtime() -> str:
    """liblabrador_ldpc_hip.so is linked without the HIP runtime; supply ONE for the process.

    If torch is installed, its bundled libamdhip64.so is used (found without importing torch), so
    that this library and torch share one runtime -- device pointers, streams and events are then
    interchangeable, whichever is imported first.  Otherwise the system ROCm runtime is used."""
    import importlib.util
    cands = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            return c
        except OSError as e:      # try the next candidate
            errs.append(f"{c}: {e}")
    raise ImportError("no HIP runtime (libamdhip64.so) could be loaded: " + "; ".join(errs))


def _load() -> ctypes.CDLL:
    global LIB_PATH
    LIB_PATH = os.environ.get("LABRADOR_LDPC_HIP_LIB", LIB_PATH)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C labrador_ldpc_amd/csrc -j8`.  The HIP library is required (no CPU fallback).")
    global HIP_RUNTIME
    HIP_RUNTIME = _preload_hip_runtime()
    dll = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(dll, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    return dll


HIP_RUNTIME = ""
lib = _load()


def device_count() -> int:
    return int(lib.labrador_ldpc_hip_device_count())


def last_error() -> str:
    return lib.labrador_ldpc_hip_last_error().decode()


def _check(status: int) -> None:
    if status != 0:
        raise LdpcHipError(f"status {status}: {last_error()}")


_NP_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.int8): "i8", np.dtype(np.int16): "i16",
              np.dtype(np.int32): "i32", np.dtype(np.float64): "f64"}


_DECODE_MS_FN: dict = {}
_SIZES: dict = {}


def _sizes(code) -> Tuple[int, int]:
    """(n, output_len) of a code, asked of the library once"""
    v = _SIZES.get(code)
    if v is None:
        v = _SIZES[code] = (code.n(), code.output_len())
    return v


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(a) -> int:
    return a.data_ptr() if _is_torch(a) else a.ctypes.data


def _suffix(a) -> str:
    if _is_torch(a):
        import torch
        return {torch.float32: "f32", torch.int8: "i8", torch.int16: "i16", torch.int32: "i32", torch.float64: "f64"}[a.dtype]
    return _NP_SUFFIX[a.dtype]


def _host_opts(stream, variant: int, devices):
    """opts for host (numpy) buffers.  `devices`: None = the current device, "all" = every gfx950
    device, or a sequence of HIP ordinals (may repeat) to shard the batch over."""
    if devices is None:
        return HipOpts(DEVICE_CURRENT, MEM_HOST, stream, variant, 0, None), None
    if stream is not None:
        raise ValueError("a device set runs on the library's own streams: stream must be None")
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError('devices must be None, "all" or a sequence of device ordinals')
        return HipOpts(DEVICE_ALL, MEM_HOST, None, variant, 0, None), None
    devs = [int(d) for d in devices]
    if not devs:
        raise ValueError("empty device list")
    arr = (ctypes.c_int * len(devs))(*devs)
    return HipOpts(DEVICE_CURRENT, MEM_HOST, None, variant, len(devs), arr), arr   # keep arr alive during the call


def _check_result_buffer(buf, like, shape, kinds, name: str):
    """A caller-supplied result buffer is handed to the C ABI as a raw pointer: refuse anything the
    kernel or the copy-out would overrun or misinterpret."""
    if _is_torch(like):
        import torch
        ok_dtypes = {"u8": (torch.uint8,), "i32": (torch.int32, getattr(torch, "uint32", torch.int32))}[kinds]
        if not _is_torch(buf) or buf.device != like.device:
            raise ValueError(f"{name} must be a torch tensor on {like.device}")
        if buf.dtype not in ok_dtypes or tuple(buf.shape) != shape or not buf.is_contiguous():
            raise ValueError(f"{name} must be a contiguous {kinds} tensor of shape {shape}")
    else:
        ok_dtypes = {"u8": (np.uint8,), "i32": (np.uint32, np.int32)}[kinds]
        if not isinstance(buf, np.ndarray) or buf.dtype not in [np.dtype(d) for d in ok_dtypes]:
            raise ValueError(f"{name} must be a numpy array of dtype {kinds}")
        if buf.shape != shape or not buf.flags.c_contiguous or not buf.flags.writeable:
            raise ValueError(f"{name} must be a writable C-contiguous array of shape {shape}")


# (submatrix_size, circulant_size) per code: src/codes/mod.rs:109-241
_SUBMATRIX = {0: (16, 16), 1: (32, 32), 2: (64, 64), 3: (128, 32), 4: (256, 64), 5: (512, 128),
              6: (512, 128), 7: (1024, 256), 8: (2048, 512)}


class LDPCCode(enum.IntEnum):
    """`enum LDPCCode` (src/codes/mod.rs:37-66) with the crate's method names."""

    TC128 = 0
    TC256 = 1
    TC512 = 2
    TM1280 = 3
    TM1536 = 4
    TM2048 = 5
    TM5120 = 6
    TM6144 = 7
    TM8192 = 8

    # ---- parameters: src/codes/mod.rs:381-409 ----
    def n(self) -> int:
        return int(lib.labrador_ldpc_code_n(int(self)))

    def k(self) -> int:
        return int(lib.labrador_ldpc_code_k(int(self)))

    def punctured_bits(self) -> int:
        return self.output_len() * 8 - self.n()

    def submatrix_size(self) -> int:
        return _SUBMATRIX[int(self)][0]

    def circulant_size(self) -> int:
        return _SUBMATRIX[int(self)][1]

    def paritycheck_sum(self) -> int:
        return (self.decode_ms_working_len() - 3 * self.n() - 3 * self.punctured_bits() + 2 * self.k()) // 2

    def iter_paritychecks(self):
        """(check, variable) index arrays of every parity-check edge in the crate's iteration order
        (src/codes/mod.rs:435-441), from the tables the kernels are generated from."""
        E = self.paritycheck_sum()
        chk, var = np.empty(E, dtype=np.uint16), np.empty(E, dtype=np.uint16)
        got = lib.labrador_ldpc_hip_edges(int(self), chk.ctypes.data, var.ctypes.data, E)
        assert got == E
        return chk, var

    # ---- sizes: src/decoder.rs:93-116 ----
    def decode_bf_working_len(self) -> int:
        return int(lib.labrador_ldpc_bf_working_len(int(self)))

    def decode_ms_working_len(self) -> int:
        return int(lib.labrador_ldpc_ms_working_len(int(self)))

    def decode_ms_working_u8_len(self) -> int:
        return int(lib.labrador_ldpc_ms_working_u8_len(int(self)))

    def output_len(self) -> int:
        return int(lib.labrador_ldpc_output_len(int(self)))

    # ---- encoder: src/encoder.rs:293-315 ----
    def encode(self, codeword: np.ndarray) -> np.ndarray:
        """Set the parity bytes of `codeword` (n/8 bytes, first k/8 = data) in place."""
        cw = _as_u8(codeword, self.n() // 8, "codeword must be n bits long")
        lib.labrador_ldpc_encode(int(self), cw.ctypes.data)
        return cw

    def copy_encode(self, data: np.ndarray, codeword: np.ndarray) -> np.ndarray:
        d = _as_u8(data, self.k() // 8, "data must be k bits long")
        cw = _as_u8(codeword, self.n() // 8, "codeword must be n bits long")
        lib.labrador_ldpc_copy_encode(int(self), d.ctypes.data, cw.ctypes.data)
        return cw

    # ---- LLR helpers: src/decoder.rs:484-509 ----
    def hard_to_llrs(self, input: np.ndarray, llrs: np.ndarray) -> None:
        inp = _as_u8(input, self.n() // 8, "input.len() != n/8")
        if llrs.shape != (self.n(),):
            raise ValueError("llrs.len() != n")
        getattr(lib, "labrador_ldpc_hard_to_llrs_" + _suffix(llrs))(int(self), inp.ctypes.data, llrs.ctypes.data)

    def llrs_to_hard(self, llrs: np.ndarray, output: np.ndarray) -> None:
        if llrs.shape != (self.n(),):
            raise ValueError("llrs.len() != n")
        out = _as_u8(output, self.n() // 8, "output.len() != n/8")
        getattr(lib, "labrador_ldpc_llrs_to_hard_" + _suffix(llrs))(int(self), llrs.ctypes.data, out.ctypes.data)

    # ---- min-sum decoder: src/decoder.rs:347-475 ----
    def decode_ms(self, llrs: np.ndarray, output: np.ndarray, working: Optional[np.ndarray] = None,
                  working_u8: Optional[np.ndarray] = None, maxiters: int = 50) -> Tuple[bool, int]:
        """One codeword on the GPU.  Returns (success, iterations) like the crate.

        Length checks mirror the asserts of src/decoder.rs:356-359; `working`/`working_u8`
        are optional here (the GPU keeps all message state on chip) but are length-checked
        when given."""
        if not isinstance(llrs, np.ndarray) or llrs.dtype not in _NP_SUFFIX:
            raise ValueError("llrs must be a numpy array of dtype int8, int16, int32, float32 or float64")
        # (a call is ~12 us in the library: the sizes come from a per-code cache and the addresses from __array_interface__, not
        # from three more trips through ctypes and two `.ctypes` objects -- 17 -> 14 us per call through this wrapper)
        n, out_len = _sizes(self)
        if llrs.shape != (n,):
            raise ValueError("llrs.len() != n")
        out = _as_u8(output, out_len, "output.len() != (n+p)/8")
        if working is not None and working.shape != (self.decode_ms_working_len(),):
            raise ValueError("working.len() incorrect")
        if working_u8 is not None and working_u8.shape != (self.decode_ms_working_u8_len(),):
            raise ValueError("working_u8 != (n+p-k)/8")
        if not llrs.flags.c_contiguous:
            llrs = np.ascontiguousarray(llrs)
        iters = ctypes.c_size_t(0)
        fn = _DECODE_MS_FN.get(llrs.dtype)
        if fn is None:
            fn = _DECODE_MS_FN[llrs.dtype] = getattr(lib, "labrador_ldpc_decode_ms_" + _NP_SUFFIX[llrs.dtype])
        ok = fn(int(self), llrs.__array_interface__["data"][0], out.__array_interface__["data"][0],
                working.ctypes.data if working is not None else None,
                working_u8.ctypes.data if working_u8 is not None else None,
                maxiters, ctypes.byref(iters))
        if not ok:
            err = last_error()
            if err:
                raise LdpcHipError(err)
        return bool(ok), int(iters.value)

    def decode_ms_batch(self, llrs, maxiters: int = 50, output=None, iters=None, success=None,
                        variant: int = 0, stream: Optional[int] = None, devices=None):
        """Decode `llrs[batch, n]`.

        numpy arrays are host buffers (the call stages them and returns when results are
        back; `devices="all"` or a list of HIP ordinals shards the batch over several GPUs);
        torch CUDA tensors are device-resident buffers: the call only enqueues the
        kernel on the tensor's device, on `stream` (default: torch's current stream).
        Returns (output[batch, output_len] u8, iters[batch] u32/i32, success[batch] u8)."""
        if not (_is_torch(llrs) or isinstance(llrs, np.ndarray)):
            raise ValueError("llrs must be a numpy array (host) or a torch CUDA tensor (device)")
        if llrs.ndim != 2 or llrs.shape[1] != self.n():
            raise ValueError("llrs must be [batch, n]")
        batch = llrs.shape[0]
        try:
            fn = getattr(lib, "labrador_ldpc_decode_ms_batch_" + _suffix(llrs), None)
        except KeyError:
            fn = None
        if fn is None:
            raise LdpcHipError(f"no batched kernel for dtype {llrs.dtype}")
        keep = None
        if _is_torch(llrs):
            import torch
            if not llrs.is_cuda:
                raise ValueError("torch tensors must live on the GPU (use numpy for host buffers)")
            if not llrs.is_contiguous():
                raise ValueError("llrs must be contiguous")
            if devices is not None:
                raise ValueError("device-resident buffers live on one device; `devices` is for host buffers")
            dev = llrs.device
            if output is None:
                output = torch.empty((batch, self.output_len()), dtype=torch.uint8, device=dev)
            if iters is None:
                iters = torch.empty((batch,), dtype=torch.int32, device=dev)
            if success is None:
                success = torch.empty((batch,), dtype=torch.uint8, device=dev)
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, variant, 0, None)
        else:
            llrs = np.ascontiguousarray(llrs)
            if output is None:
                output = np.empty((batch, self.output_len()), dtype=np.uint8)
            if iters is None:
                iters = np.empty((batch,), dtype=np.uint32)
            if success is None:
                success = np.empty((batch,), dtype=np.uint8)
            opts, keep = _host_opts(stream, variant, devices)
        _check_result_buffer(output, llrs, (batch, self.output_len()), "u8", "output")
        _check_result_buffer(iters, llrs, (batch,), "i32", "iters")
        _check_result_buffer(success, llrs, (batch,), "u8", "success")
        _check(fn(int(self), _ptr(llrs), _ptr(output), _ptr(iters), _ptr(success), batch, maxiters,
                  ctypes.byref(opts)))
        del keep
        return output, iters, success

    def decode_ms_batch_multi(self, parts, maxiters: int = 50, variant: int = 0):
        """Decode several device-resident batches -- one torch CUDA tensor `llrs[frames_i, n]` per part, each on its own (or the
        same) GPU -- with ONE call: every part is enqueued by the library's worker of its device and the call returns when all
        results are in place (labrador_ldpc_decode_ms_batch_*_multi; the reference's harness shape, one job over all workers,
        perftest/src/main.rs:39-52).  Returns a list of (output, iters, success) tensors, one triple per part."""
        import torch
        parts = list(parts)
        if not parts:
            return []
        if not all(_is_torch(p) and p.is_cuda and p.is_contiguous() and p.ndim == 2 and p.shape[1] == self.n() for p in parts):
            raise ValueError("every part must be a contiguous torch CUDA tensor [frames, n]")
        if len({p.dtype for p in parts}) != 1:
            raise ValueError("all parts must have one LLR type")
        fn = getattr(lib, "labrador_ldpc_decode_ms_batch_" + _suffix(parts[0]) + "_multi")
        res = [(torch.empty((p.shape[0], self.output_len()), dtype=torch.uint8, device=p.device),
                torch.empty((p.shape[0],), dtype=torch.int32, device=p.device),
                torch.empty((p.shape[0],), dtype=torch.uint8, device=p.device)) for p in parts]
        # The library's streams know nothing of torch's: what produced the inputs must be done.  Only torch's CURRENT stream of each
        # device is waited for here -- a caller that filled a part on another stream synchronises that stream itself before the call.
        for p in parts:
            torch.cuda.current_stream(p.device).synchronize()
        n = len(parts)
        arr = lambda vals, t: (t * n)(*vals)
        # get_device(): the ordinal the memory lives on, also for a tensor made on torch.device("cuda") (index None) while another
        # GPU than 0 is current (round 5 advice: `index or 0` sent such a part to device 0)
        devs = arr([p.get_device() for p in parts], ctypes.c_int)
        _check(fn(int(self), n, devs, arr([_ptr(p) for p in parts], ctypes.c_void_p), arr([_ptr(r[0]) for r in res], ctypes.c_void_p),
                  arr([_ptr(r[1]) for r in res], ctypes.c_void_p), arr([_ptr(r[2]) for r in res], ctypes.c_void_p),
                  arr([p.shape[0] for p in parts], ctypes.c_size_t), maxiters, variant))
        return res

    # ---- bit-flipping decoder: src/decoder.rs:243-301 ----
    def decode_bf(self, input: np.ndarray, output: np.ndarray, working: Optional[np.ndarray] = None,
                  maxiters: int = 50) -> Tuple[bool, int]:
        """One codeword on the GPU; (success, iterations) like the crate (length checks of :247-249)."""
        inp = _as_u8(input, self.n() // 8, "input.len() != n/8")
        out = _as_u8(output, self.output_len(), "output.len != (n+p)/8")
        if working is not None and working.shape != (self.decode_bf_working_len(),):
            raise ValueError("working.len() incorrect")
        iters = ctypes.c_size_t(0)
        ok = lib.labrador_ldpc_decode_bf(int(self), inp.ctypes.data, out.ctypes.data, None, maxiters, ctypes.byref(iters))
        err = last_error()
        if not ok and err:
            raise LdpcHipError(err)
        return bool(ok), int(iters.value)

    def decode_bf_batch(self, input, maxiters: int = 50, stream: Optional[int] = None, devices=None):
        """input[batch, n/8] -> (output[batch, output_len], iters[batch], success[batch]).
        numpy = host buffers, torch CUDA uint8 tensors = device buffers (asynchronous)."""
        if input.ndim != 2 or input.shape[1] != self.n() // 8:
            raise ValueError("input must be [batch, n/8]")
        batch = input.shape[0]
        if _is_torch(input):
            import torch
            if not (input.is_cuda and input.dtype == torch.uint8 and input.is_contiguous()):
                raise ValueError("input must be a contiguous uint8 CUDA tensor")
            dev = input.device
            output = torch.empty((batch, self.output_len()), dtype=torch.uint8, device=dev)
            iters = torch.empty((batch,), dtype=torch.int32, device=dev)
            success = torch.empty((batch,), dtype=torch.uint8, device=dev)
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, 0, 0, None)
        else:
            input = np.ascontiguousarray(input, dtype=np.uint8)
            output = np.empty((batch, self.output_len()), dtype=np.uint8)
            iters = np.empty((batch,), dtype=np.uint32)
            success = np.empty((batch,), dtype=np.uint8)
            opts, _keep = _host_opts(stream, 0, devices)
        _check(lib.labrador_ldpc_decode_bf_batch(int(self), _ptr(input), _ptr(output), _ptr(iters), _ptr(success),
                                                 batch, maxiters, ctypes.byref(opts)))
        return output, iters, success

    def encode_batch(self, data, codewords=None, stream: Optional[int] = None, devices=None):
        """Batched `copy_encode`: data[batch, k/8] -> codewords[batch, n/8] on the GPU.
        numpy = host buffers (synchronous), torch CUDA uint8 tensors = device buffers (asynchronous)."""
        if data.ndim != 2 or data.shape[1] != self.k() // 8:
            raise ValueError("data must be [batch, k/8]")
        batch = data.shape[0]
        if _is_torch(data):
            import torch
            if not (data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()):
                raise ValueError("data must be a contiguous uint8 CUDA tensor")
            dev = data.device
            if codewords is None:
                codewords = torch.empty((batch, self.n() // 8), dtype=torch.uint8, device=dev)
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, 0, 0, None)
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            if codewords is None:
                codewords = np.empty((batch, self.n() // 8), dtype=np.uint8)
            opts, _keep = _host_opts(stream, 0, devices)
        _check_result_buffer(codewords, data, (batch, self.n() // 8), "u8", "codewords")
        _check(lib.labrador_ldpc_encode_batch(int(self), _ptr(data), _ptr(codewords), batch, ctypes.byref(opts)))
        return codewords

    # ---- LLR helpers, batched (src/decoder.rs:484-509 frame after frame) ----
    def hard_to_llrs_batch(self, input, dtype="f32", llrs=None, stream: Optional[int] = None):
        """input[batch, n/8] packed bits -> llrs[batch, n] of +-1 (`dtype`: "i8", "i16", "i32", "f32", "f64").
        torch CUDA uint8 tensor = on the device, asynchronous on the stream; numpy = host code."""
        if input.ndim != 2 or input.shape[1] != self.n() // 8:
            raise ValueError("input must be [batch, n/8]")
        batch = input.shape[0]
        if _is_torch(input):
            import torch
            tdt = {"f32": torch.float32, "i8": torch.int8, "i16": torch.int16, "i32": torch.int32, "f64": torch.float64}[dtype]
            if not (input.is_cuda and input.dtype == torch.uint8 and input.is_contiguous()):
                raise ValueError("input must be a contiguous uint8 CUDA tensor")
            dev = input.device
            if llrs is None:
                llrs = torch.empty((batch, self.n()), dtype=tdt, device=dev)
            elif not (_is_torch(llrs) and llrs.device == dev and llrs.dtype == tdt and tuple(llrs.shape) == (batch, self.n())
                      and llrs.is_contiguous()):
                raise ValueError(f"llrs must be a contiguous {tdt} tensor of shape ({batch}, {self.n()}) on {dev}")
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, 0, 0, None)
        else:
            ndt = {v: k for k, v in _NP_SUFFIX.items()}[dtype]
            input = np.ascontiguousarray(input, dtype=np.uint8)
            if llrs is None:
                llrs = np.empty((batch, self.n()), dtype=ndt)
            elif not (isinstance(llrs, np.ndarray) and llrs.dtype == ndt and llrs.shape == (batch, self.n())
                      and llrs.flags.c_contiguous and llrs.flags.writeable):
                raise ValueError(f"llrs must be a writable C-contiguous {ndt} array of shape ({batch}, {self.n()})")
            opts = HipOpts(DEVICE_CURRENT, MEM_HOST, None, 0, 0, None)
        _check(getattr(lib, "labrador_ldpc_hard_to_llrs_batch_" + dtype)(int(self), _ptr(input), _ptr(llrs), batch, ctypes.byref(opts)))
        return llrs

    def llrs_to_hard_batch(self, llrs, output=None, stream: Optional[int] = None):
        """llrs[batch, n] -> output[batch, n/8] packed hard decisions (bit set where the LLR is < 0)."""
        if llrs.ndim != 2 or llrs.shape[1] != self.n():
            raise ValueError("llrs must be [batch, n]")
        batch = llrs.shape[0]
        if _is_torch(llrs):
            import torch
            if not (llrs.is_cuda and llrs.is_contiguous()):
                raise ValueError("llrs must be a contiguous CUDA tensor")
            dev = llrs.device
            if output is None:
                output = torch.empty((batch, self.n() // 8), dtype=torch.uint8, device=dev)
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, 0, 0, None)
        else:
            if llrs.dtype not in _NP_SUFFIX:
                raise ValueError("llrs dtype must be one of int8, int16, int32, float32, float64")
            llrs = np.ascontiguousarray(llrs)
            if output is None:
                output = np.empty((batch, self.n() // 8), dtype=np.uint8)
            opts = HipOpts(DEVICE_CURRENT, MEM_HOST, None, 0, 0, None)
        _check_result_buffer(output, llrs, (batch, self.n() // 8), "u8", "output")
        _check(getattr(lib, "labrador_ldpc_llrs_to_hard_batch_" + _suffix(llrs))(int(self), _ptr(llrs), _ptr(output), batch, ctypes.byref(opts)))
        return output

    # ---- synthetic channel (harness) ----
    def awgn_frames(self, codewords, batch: int, sigma: float, seed: int, dtype="f32",
                    scale: float = 8.0, lim: int = 31, out=None, stream: Optional[int] = None, first_frame: int = 0):
        """Fill `out[batch, n]` (device tensor) with BPSK+AWGN LLRs of the device-resident
        codeword pool `codewords[pool, n/8]` (see labrador_ldpc_hip_awgn_*_at): frames
        [first_frame, first_frame + batch) of the job `seed` names -- a shard generated with its own
        first_frame equals that slice of the whole job's buffer byte for byte."""
        import torch
        if not (codewords.is_cuda and codewords.dtype == torch.uint8 and codewords.is_contiguous()):
            raise ValueError("codewords must be a contiguous uint8 CUDA tensor [pool, n/8]")
        dev = codewords.device
        tdt = {"f32": torch.float32, "i8": torch.int8}[dtype]
        if out is None:
            out = torch.empty((batch, self.n()), dtype=tdt, device=dev)
        elif not (_is_torch(out) and out.device == dev and out.dtype == tdt and tuple(out.shape) == (batch, self.n())
                  and out.is_contiguous()):
            raise ValueError(f"out must be a contiguous {tdt} tensor of shape ({batch}, {self.n()}) on {dev}")
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        opts = HipOpts(dev.index if dev.index is not None else -1, MEM_DEVICE, stream, 0, 0, None)
        if dtype == "f32":
            _check(lib.labrador_ldpc_hip_awgn_f32_at(int(self), codewords.data_ptr(), codewords.shape[0],
                                                     out.data_ptr(), first_frame, batch, sigma, seed, ctypes.byref(opts)))
        else:
            _check(lib.labrador_ldpc_hip_awgn_i8_at(int(self), codewords.data_ptr(), codewords.shape[0],
                                                    out.data_ptr(), first_frame, batch, sigma, scale, lim, seed,
                                                    ctypes.byref(opts)))
        return out


def _as_u8(a: np.ndarray, length: int, msg: str) -> np.ndarray:
    if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 1 or not a.flags.c_contiguous:
        raise ValueError("expected a contiguous 1-D uint8 numpy array")
    if a.shape[0] != length:
        raise ValueError(msg)
    return a
