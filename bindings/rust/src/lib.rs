//! Rust host binding of `liblabrador_ldpc_hip.so` (include/labrador_ldpc_hip.h, ABI 3).
//!
//! The reference crate exports its decoder to C in `capi/src/lib.rs:15-179`; this file is the same boundary walked the other
//! way: a Rust host (the crate itself behind a `hip` feature, or its `perftest` harness) calls the MI355X library through
//! `extern "C"`.  `LDPCCode` is `#[repr(C)]` with the reference's discriminants (src/codes/mod.rs:37-66), so it passes by
//! value as `enum labrador_ldpc_code`, exactly as it does in the reference's own C API.
//!
//! The safe wrappers keep the crate's surface for this path -- `decode_ms::<T>` with the same arguments, asserts and
//! `(bool, usize)` result (src/decoder.rs:347-475), `hard_to_llrs`, `llrs_to_hard`, the length functions -- and add the
//! batched calls the GPU path sits behind.  There is no CPU fallback: when the library cannot run, the single-frame calls
//! return `(false, maxiters)` with zeroed output and `last_error()` says why.
//!
//! Not compiled in this repository's image (no Rust toolchain); `tests/test_rust_shim.py` checks every declaration of the
//! `extern "C"` block against the header.
#![allow(non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]

use core::ffi::{c_char, c_int, c_void};

/// `enum labrador_ldpc_code` (capi/include/labrador_ldpc.h:19-29; src/codes/mod.rs:37-66).
#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq, Eq, Hash)]
pub enum LDPCCode {
    TC128 = 0,
    TC256 = 1,
    TC512 = 2,
    TM1280 = 3,
    TM1536 = 4,
    TM2048 = 5,
    TM5120 = 6,
    TM6144 = 7,
    TM8192 = 8,
}

/// `LABRADOR_LDPC_HIP_ABI`: the layout of [`HipOpts`] this file was written against.
pub const ABI: c_int = 3;

pub const OK: c_int = 0;
pub const EINVAL: c_int = -1;
pub const ENODEV: c_int = -2;
pub const ERUNTIME: c_int = -3;
pub const EUNSUPPORTED: c_int = -4;
pub const MEM_HOST: c_int = 0;
pub const MEM_DEVICE: c_int = 1;
pub const DEVICE_CURRENT: c_int = -1;
/// Host batches only: every gfx950 device, one worker thread and copy/kernel/copy pipeline each.
pub const DEVICE_ALL: c_int = -2;
/// `opts.variant` flag: fixed-stride distribution of the codewords instead of the launch's queue.
pub const VARIANT_STATIC: c_int = 256;
/// `opts.variant` flags: force the one-launch / the two-launch handling of NaN LLRs on the kernel that has both
/// (TM5120 f32, TM1280 f32; by default the batch size decides).  Results are identical.
pub const VARIANT_ONE_PASS: c_int = 512;
pub const VARIANT_TWO_PASS: c_int = 1024;

/// `struct labrador_ldpc_hip_opts`.  `struct_size` makes the struct growable: the library reads a field only if it lies
/// inside the first `struct_size` bytes.  Use [`HipOpts::new`].
#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct HipOpts {
    pub struct_size: usize,
    pub device: c_int,
    pub memory: c_int,
    pub stream: *mut c_void,
    pub variant: c_int,
    pub n_devices: c_int,
    pub devices: *const c_int,
}

impl HipOpts {
    /// Host buffers, the calling thread's current device, default stream, tuned kernel.
    pub fn new() -> Self {
        HipOpts {
            struct_size: core::mem::size_of::<HipOpts>(),
            device: DEVICE_CURRENT,
            memory: MEM_HOST,
            stream: core::ptr::null_mut(),
            variant: 0,
            n_devices: 0,
            devices: core::ptr::null(),
        }
    }
}

impl Default for HipOpts {
    fn default() -> Self {
        Self::new()
    }
}

extern "C" {
    // ---- Part 1: the reference's 21 symbols, same signatures (capi/src/lib.rs:15-179) -------------------------------
    pub fn labrador_ldpc_code_n(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_code_k(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_bf_working_len(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_ms_working_len(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_ms_working_u8_len(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_output_len(code: LDPCCode) -> usize;
    pub fn labrador_ldpc_encode(code: LDPCCode, codeword: *mut u8);
    pub fn labrador_ldpc_copy_encode(code: LDPCCode, data: *const u8, codeword: *mut u8);
    pub fn labrador_ldpc_decode_bf(code: LDPCCode, input: *const u8, output: *mut u8, working: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_decode_ms_i8(code: LDPCCode, llrs: *const i8, output: *mut u8, working: *mut i8, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_decode_ms_i16(code: LDPCCode, llrs: *const i16, output: *mut u8, working: *mut i16, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_decode_ms_f32(code: LDPCCode, llrs: *const f32, output: *mut u8, working: *mut f32, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_decode_ms_f64(code: LDPCCode, llrs: *const f64, output: *mut u8, working: *mut f64, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_hard_to_llrs_i8(code: LDPCCode, input: *const u8, llrs: *mut i8);
    pub fn labrador_ldpc_hard_to_llrs_i16(code: LDPCCode, input: *const u8, llrs: *mut i16);
    pub fn labrador_ldpc_hard_to_llrs_f32(code: LDPCCode, input: *const u8, llrs: *mut f32);
    pub fn labrador_ldpc_hard_to_llrs_f64(code: LDPCCode, input: *const u8, llrs: *mut f64);
    pub fn labrador_ldpc_llrs_to_hard_i8(code: LDPCCode, llrs: *const i8, output: *mut u8);
    pub fn labrador_ldpc_llrs_to_hard_i16(code: LDPCCode, llrs: *const i16, output: *mut u8);
    pub fn labrador_ldpc_llrs_to_hard_f32(code: LDPCCode, llrs: *const f32, output: *mut u8);
    pub fn labrador_ldpc_llrs_to_hard_f64(code: LDPCCode, llrs: *const f64, output: *mut u8);
    // ---- the i32 forms the crate's generic has but its C API lacks (src/decoder.rs:60-68) -----------------------------
    pub fn labrador_ldpc_decode_ms_i32(code: LDPCCode, llrs: *const i32, output: *mut u8, working: *mut i32, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    pub fn labrador_ldpc_hard_to_llrs_i32(code: LDPCCode, input: *const u8, llrs: *mut i32);
    pub fn labrador_ldpc_llrs_to_hard_i32(code: LDPCCode, llrs: *const i32, output: *mut u8);
    // ---- Part 2: batched GPU entry points -------------------------------------------------------------------------------
    pub fn labrador_ldpc_decode_ms_batch_f32(code: LDPCCode, llrs: *const f32, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i8(code: LDPCCode, llrs: *const i8, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i16(code: LDPCCode, llrs: *const i16, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i32(code: LDPCCode, llrs: *const i32, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_f64(code: LDPCCode, llrs: *const f64, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_f32_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const f32, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i8_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const i8, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i16_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const i16, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_i32_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const i32, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    pub fn labrador_ldpc_decode_ms_batch_f64_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const f64, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    pub fn labrador_ldpc_decode_bf_batch(code: LDPCCode, input: *const u8, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_encode_batch(code: LDPCCode, data: *const u8, codewords: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hard_to_llrs_batch_i8(code: LDPCCode, input: *const u8, llrs: *mut i8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hard_to_llrs_batch_i16(code: LDPCCode, input: *const u8, llrs: *mut i16, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hard_to_llrs_batch_i32(code: LDPCCode, input: *const u8, llrs: *mut i32, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hard_to_llrs_batch_f32(code: LDPCCode, input: *const u8, llrs: *mut f32, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hard_to_llrs_batch_f64(code: LDPCCode, input: *const u8, llrs: *mut f64, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_llrs_to_hard_batch_i8(code: LDPCCode, llrs: *const i8, output: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_llrs_to_hard_batch_i16(code: LDPCCode, llrs: *const i16, output: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_llrs_to_hard_batch_i32(code: LDPCCode, llrs: *const i32, output: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_llrs_to_hard_batch_f32(code: LDPCCode, llrs: *const f32, output: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_llrs_to_hard_batch_f64(code: LDPCCode, llrs: *const f64, output: *mut u8, batch: usize, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hip_awgn_f32(code: LDPCCode, codewords: *const u8, pool: usize, llrs: *mut f32, batch: usize, sigma: f32, seed: u64, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hip_awgn_i8(code: LDPCCode, codewords: *const u8, pool: usize, llrs: *mut i8, batch: usize, sigma: f32, scale: f32, lim: c_int, seed: u64, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hip_awgn_f32_at(code: LDPCCode, codewords: *const u8, pool: usize, llrs: *mut f32, first_frame: u64, batch: usize, sigma: f32, seed: u64, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hip_awgn_i8_at(code: LDPCCode, codewords: *const u8, pool: usize, llrs: *mut i8, first_frame: u64, batch: usize, sigma: f32, scale: f32, lim: c_int, seed: u64, opts: *const HipOpts) -> c_int;
    pub fn labrador_ldpc_hip_edge_crc(code: LDPCCode) -> u32;
    pub fn labrador_ldpc_hip_edges(code: LDPCCode, checks: *mut u16, variables: *mut u16, cap: usize) -> usize;
    pub fn labrador_ldpc_hip_shard_range(batch: usize, parts: usize, index: usize, first: *mut usize, count: *mut usize) -> c_int;
    pub fn labrador_ldpc_hip_device_count() -> c_int;
    pub fn labrador_ldpc_hip_last_error() -> *const c_char;
    pub fn labrador_ldpc_hip_version() -> *const c_char;
    pub fn labrador_ldpc_hip_build_id() -> *const c_char;
    pub fn labrador_ldpc_hip_shader_clock_mhz(device: c_int, busy_ms: f64, mhz: *mut f64) -> c_int;
    pub fn labrador_ldpc_hip_abi_version() -> c_int;
    pub fn labrador_ldpc_hip_decode_ms_i8_kernel(code: LDPCCode, variant: c_int, batch: usize) -> *const c_char;
}

/// The calling thread's last failure ("" if none).
pub fn last_error() -> String {
    unsafe { core::ffi::CStr::from_ptr(labrador_ldpc_hip_last_error()).to_string_lossy().into_owned() }
}

/// `true` if the loaded library speaks the ABI this file was written against.
pub fn abi_matches() -> bool {
    unsafe { labrador_ldpc_hip_abi_version() == ABI }
}

/// The LLR types of the crate's `DecodeFrom` trait (src/decoder.rs:22-86), each bound to its C entry points.
pub trait DecodeFrom: Copy {
    unsafe fn decode_ms(code: LDPCCode, llrs: *const Self, output: *mut u8, working: *mut Self, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool;
    unsafe fn decode_ms_batch(code: LDPCCode, llrs: *const Self, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int;
    unsafe fn decode_ms_batch_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const Self, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int;
    unsafe fn hard_to_llrs(code: LDPCCode, input: *const u8, llrs: *mut Self);
    unsafe fn llrs_to_hard(code: LDPCCode, llrs: *const Self, output: *mut u8);
}

/// One device-resident batch of a multi-GPU decode: `frames` frames whose four buffers live on HIP device `device`.
pub struct DevicePart<T> {
    pub device: c_int,
    pub llrs: *const T,
    pub output: *mut u8,
    pub iters: *mut u32,
    pub success: *mut u8,
    pub frames: usize,
}

macro_rules! decode_from {
    ($t:ty, $ms:ident, $batch:ident, $multi:ident, $h2l:ident, $l2h:ident) => {
        impl DecodeFrom for $t {
            unsafe fn decode_ms(code: LDPCCode, llrs: *const Self, output: *mut u8, working: *mut Self, working_u8: *mut u8, max_iters: usize, iters_run: *mut usize) -> bool {
                $ms(code, llrs, output, working, working_u8, max_iters, iters_run)
            }
            unsafe fn decode_ms_batch(code: LDPCCode, llrs: *const Self, output: *mut u8, iters: *mut u32, success: *mut u8, batch: usize, max_iters: usize, opts: *const HipOpts) -> c_int {
                $batch(code, llrs, output, iters, success, batch, max_iters, opts)
            }
            unsafe fn decode_ms_batch_multi(code: LDPCCode, n_parts: usize, devices: *const c_int, llrs: *const *const Self, output: *const *mut u8, iters: *const *mut u32, success: *const *mut u8, frames: *const usize, max_iters: usize, variant: c_int) -> c_int {
                $multi(code, n_parts, devices, llrs, output, iters, success, frames, max_iters, variant)
            }
            unsafe fn hard_to_llrs(code: LDPCCode, input: *const u8, llrs: *mut Self) {
                $h2l(code, input, llrs)
            }
            unsafe fn llrs_to_hard(code: LDPCCode, llrs: *const Self, output: *mut u8) {
                $l2h(code, llrs, output)
            }
        }
    };
}
decode_from!(i8, labrador_ldpc_decode_ms_i8, labrador_ldpc_decode_ms_batch_i8, labrador_ldpc_decode_ms_batch_i8_multi, labrador_ldpc_hard_to_llrs_i8, labrador_ldpc_llrs_to_hard_i8);
decode_from!(i16, labrador_ldpc_decode_ms_i16, labrador_ldpc_decode_ms_batch_i16, labrador_ldpc_decode_ms_batch_i16_multi, labrador_ldpc_hard_to_llrs_i16, labrador_ldpc_llrs_to_hard_i16);
decode_from!(i32, labrador_ldpc_decode_ms_i32, labrador_ldpc_decode_ms_batch_i32, labrador_ldpc_decode_ms_batch_i32_multi, labrador_ldpc_hard_to_llrs_i32, labrador_ldpc_llrs_to_hard_i32);
decode_from!(f32, labrador_ldpc_decode_ms_f32, labrador_ldpc_decode_ms_batch_f32, labrador_ldpc_decode_ms_batch_f32_multi, labrador_ldpc_hard_to_llrs_f32, labrador_ldpc_llrs_to_hard_f32);
decode_from!(f64, labrador_ldpc_decode_ms_f64, labrador_ldpc_decode_ms_batch_f64, labrador_ldpc_decode_ms_batch_f64_multi, labrador_ldpc_hard_to_llrs_f64, labrador_ldpc_llrs_to_hard_f64);

impl LDPCCode {
    pub fn n(self) -> usize { unsafe { labrador_ldpc_code_n(self) } }
    pub fn k(self) -> usize { unsafe { labrador_ldpc_code_k(self) } }
    pub fn output_len(self) -> usize { unsafe { labrador_ldpc_output_len(self) } }
    pub fn decode_bf_working_len(self) -> usize { unsafe { labrador_ldpc_bf_working_len(self) } }
    pub fn decode_ms_working_len(self) -> usize { unsafe { labrador_ldpc_ms_working_len(self) } }
    pub fn decode_ms_working_u8_len(self) -> usize { unsafe { labrador_ldpc_ms_working_u8_len(self) } }

    /// src/encoder.rs:293-315.
    pub fn copy_encode(self, data: &[u8], codeword: &mut [u8]) {
        assert_eq!(data.len(), self.k() / 8, "data must be k/8 long");
        assert_eq!(codeword.len(), self.n() / 8, "codeword must be n/8 long");
        unsafe { labrador_ldpc_copy_encode(self, data.as_ptr(), codeword.as_mut_ptr()) }
    }

    /// Drop-in for `decode_ms::<T>` (src/decoder.rs:347-475): same arguments, the same asserts (:356-359), the same
    /// `(success, iterations)`.  `working` / `working_u8` are accepted for source compatibility and not used: the message
    /// state lives in the GPU's registers and LDS.
    pub fn decode_ms<T: DecodeFrom>(self, llrs: &[T], output: &mut [u8], working: &mut [T], working_u8: &mut [u8], maxiters: usize) -> (bool, usize) {
        assert_eq!(llrs.len(), self.n(), "llrs.len() != n");
        assert_eq!(output.len(), self.output_len(), "output.len() != (n+p)/8");
        assert_eq!(working.len(), self.decode_ms_working_len(), "working.len() incorrect");
        assert_eq!(working_u8.len(), self.decode_ms_working_u8_len(), "working_u8 != (n+p-k)/8");
        let mut iters = 0usize;
        let ok = unsafe { T::decode_ms(self, llrs.as_ptr(), output.as_mut_ptr(), working.as_mut_ptr(), working_u8.as_mut_ptr(), maxiters, &mut iters) };
        (ok, iters)
    }

    /// src/decoder.rs:243-301 (with the erasure pre-pass).
    pub fn decode_bf(self, input: &[u8], output: &mut [u8], working: &mut [u8], maxiters: usize) -> (bool, usize) {
        assert_eq!(input.len(), self.n() / 8, "input.len() != n/8");
        assert_eq!(output.len(), self.output_len(), "output.len() != (n+p)/8");
        assert_eq!(working.len(), self.decode_bf_working_len(), "working.len() incorrect");
        let mut iters = 0usize;
        let ok = unsafe { labrador_ldpc_decode_bf(self, input.as_ptr(), output.as_mut_ptr(), working.as_mut_ptr(), maxiters, &mut iters) };
        (ok, iters)
    }

    /// src/decoder.rs:484-493.
    pub fn hard_to_llrs<T: DecodeFrom>(self, input: &[u8], llrs: &mut [T]) {
        assert_eq!(input.len(), self.n() / 8, "input.len() != n/8");
        assert_eq!(llrs.len(), self.n(), "llrs.len() != n");
        unsafe { T::hard_to_llrs(self, input.as_ptr(), llrs.as_mut_ptr()) }
    }

    /// src/decoder.rs:498-509.
    pub fn llrs_to_hard<T: DecodeFrom>(self, llrs: &[T], output: &mut [u8]) {
        assert_eq!(llrs.len(), self.n(), "llrs.len() != n");
        assert_eq!(output.len(), self.n() / 8, "output.len() != n/8");
        unsafe { T::llrs_to_hard(self, llrs.as_ptr(), output.as_mut_ptr()) }
    }

    /// Batched decode of host buffers: `llrs` is `[batch][n]`, `output` `[batch][output_len]`; per frame the three results
    /// equal `decode_ms` on that frame.  `opts = None`: the current device; `HipOpts { device: DEVICE_ALL, .. }` shards
    /// the batch over every GPU (the harness's `spawn_broadcast` over cores, perftest/src/main.rs:39-45, with GPUs for
    /// cores).  `Err(status)` on HIP / argument errors; [`last_error`] has the text.
    pub fn decode_ms_batch<T: DecodeFrom>(self, llrs: &[T], output: &mut [u8], iters: &mut [u32], success: &mut [u8], maxiters: usize, opts: Option<&HipOpts>) -> Result<(), c_int> {
        let batch = iters.len();
        assert_eq!(llrs.len(), batch * self.n(), "llrs.len() != batch * n");
        assert_eq!(output.len(), batch * self.output_len(), "output.len() != batch * (n+p)/8");
        assert_eq!(success.len(), batch, "success.len() != batch");
        if let Some(o) = opts {
            assert_eq!(o.memory, MEM_HOST, "slices are host memory; use the raw entry point for device buffers");
        }
        let o = opts.map_or(core::ptr::null(), |o| o as *const HipOpts);
        let st = unsafe { T::decode_ms_batch(self, llrs.as_ptr(), output.as_mut_ptr(), iters.as_mut_ptr(), success.as_mut_ptr(), batch, maxiters, o) };
        if st == OK { Ok(()) } else { Err(st) }
    }

    /// Device-resident batches on several GPUs with ONE call (the reference harness's shape: one job over all workers,
    /// perftest/src/main.rs:39-52): every part is enqueued by the library's worker of its device; returns when all are decoded.
    /// Unsafe: the pointers are device memory the caller vouches for.
    pub unsafe fn decode_ms_batch_multi<T: DecodeFrom>(self, parts: &[DevicePart<T>], maxiters: usize, variant: c_int) -> Result<(), c_int> {
        let devices: Vec<c_int> = parts.iter().map(|p| p.device).collect();
        let llrs: Vec<*const T> = parts.iter().map(|p| p.llrs).collect();
        let output: Vec<*mut u8> = parts.iter().map(|p| p.output).collect();
        let iters: Vec<*mut u32> = parts.iter().map(|p| p.iters).collect();
        let success: Vec<*mut u8> = parts.iter().map(|p| p.success).collect();
        let frames: Vec<usize> = parts.iter().map(|p| p.frames).collect();
        let st = T::decode_ms_batch_multi(self, parts.len(), devices.as_ptr(), llrs.as_ptr(), output.as_ptr(), iters.as_ptr(), success.as_ptr(), frames.as_ptr(), maxiters, variant);
        if st == OK { Ok(()) } else { Err(st) }
    }

    /// Batched systematic encode of host buffers: `data` is `[batch][k/8]`, `codewords` `[batch][n/8]`.
    pub fn encode_batch(self, data: &[u8], codewords: &mut [u8], opts: Option<&HipOpts>) -> Result<(), c_int> {
        let batch = data.len() / (self.k() / 8);
        assert_eq!(data.len(), batch * self.k() / 8, "data.len() is not a multiple of k/8");
        assert_eq!(codewords.len(), batch * self.n() / 8, "codewords.len() != batch * n/8");
        let o = opts.map_or(core::ptr::null(), |o| o as *const HipOpts);
        let st = unsafe { labrador_ldpc_encode_batch(self, data.as_ptr(), codewords.as_mut_ptr(), batch, o) };
        if st == OK { Ok(()) } else { Err(st) }
    }
}
