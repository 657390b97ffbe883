/* labrador_ldpc_hip.h -- C ABI of liblabrador_ldpc_hip.so, the MI355X (gfx950) min-sum decoder.
 *
 * Drop-in boundary for the decode_ms path of adamgreig/labrador-ldpc.  The first part mirrors
 * the reference's C API symbol for symbol (reference: capi/include/labrador_ldpc.h,
 * implemented in capi/src/lib.rs); the second part adds the batched entry points the GPU
 * path sits behind.  All paths below are relative to the reference repository.
 *
 * Conventions kept from the reference (capi/README.md:84-101, src/lib.rs:15-17):
 *   - the caller owns every buffer; lengths are implied by `code`, never passed;
 *   - packed bit buffers are MSB-first inside each byte (src/decoder.rs:459, :490, :506);
 *   - functions are re-entrant; concurrent calls on disjoint buffers are safe.
 * Differences, all deliberate:
 *   - an out-of-range `code` is undefined behaviour in the reference (it is the Rust enum
 *     itself, src/codes/mod.rs:37-66); here size queries return 0, decoders return false /
 *     a negative status and write nothing;
 *   - decode_ms runs on the GPU.  If no usable HIP device or kernel is available the
 *     decoders FAIL (false / negative status, message via labrador_ldpc_hip_last_error());
 *     there is no CPU fallback for the hot path;
 *   - the `working` / `working_u8` arguments of the single-codeword decoders are accepted
 *     for source compatibility and not touched (all message state lives in GPU registers/LDS).
 */
#ifndef LABRADOR_LDPC_HIP_H
#define LABRADOR_LDPC_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* capi/include/labrador_ldpc.h:19-29 == #[repr(C)] enum LDPCCode, src/codes/mod.rs:37-66 */
enum labrador_ldpc_code {
    LABRADOR_LDPC_CODE_TC128    = 0,
    LABRADOR_LDPC_CODE_TC256    = 1,
    LABRADOR_LDPC_CODE_TC512    = 2,
    LABRADOR_LDPC_CODE_TM1280   = 3,
    LABRADOR_LDPC_CODE_TM1536   = 4,
    LABRADOR_LDPC_CODE_TM2048   = 5,
    LABRADOR_LDPC_CODE_TM5120   = 6,
    LABRADOR_LDPC_CODE_TM6144   = 7,
    LABRADOR_LDPC_CODE_TM8192   = 8,
};

/* --------------------------------------------------------------------------------------
 * Compile-time sizes for static allocation (capi/include/labrador_ldpc.h:30-115; used by the
 * reference's C client capi/examples/example.c:23-37).  Every quantity Q in
 *   N  K  BF_WORKING_LEN  MS_WORKING_LEN  MS_WORKING_U8_LEN  OUTPUT_LEN
 * is available as LABRADOR_LDPC_<Q>_<code> and as LABRADOR_LDPC_<Q>(CODE), where CODE may itself
 * be a macro naming a code (two-level expansion), and LABRADOR_LDPC_CODE(CODE) gives the enum
 * constant.  Values follow src/codes/mod.rs:109-241 / src/decoder.rs:93-116:
 *   BF_WORKING_LEN = n+p, MS_WORKING_LEN = 2E+3n+3p-2k, MS_WORKING_U8_LEN = (n+p-k)/8,
 *   OUTPUT_LEN = (n+p)/8
 * and are checked against the labrador_ldpc_*_len() functions by tests/test_c_boundary.py.
 * Reference quirks: its header spells the TM6144 entries of the four *_LEN families "_TM6140"
 * (labrador_ldpc.h:76,:88,:100,:112) and gives LABRADOR_LDPC_N_TM6144 the value 6140 (:52), which
 * is wrong (n = 6144, src/codes/mod.rs:203-213).  Here the _TM6140 spellings are kept as aliases
 * so existing sources compile, _TM6144 spellings are added so the (CODE) forms work for TM6144,
 * and N_TM6144 carries the correct value 6144.
 * -------------------------------------------------------------------------------------- */
#define LABRADOR_LDPC_PASTE_(FAMILY, CODE)   FAMILY##CODE
#define LABRADOR_LDPC_CODE(CODE)              LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_CODE_, CODE)
#define LABRADOR_LDPC_N(CODE)                 LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_N_, CODE)
#define LABRADOR_LDPC_K(CODE)                 LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_K_, CODE)
#define LABRADOR_LDPC_BF_WORKING_LEN(CODE)    LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_BF_WORKING_LEN_, CODE)
#define LABRADOR_LDPC_MS_WORKING_LEN(CODE)    LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_MS_WORKING_LEN_, CODE)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN(CODE) LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_MS_WORKING_U8_LEN_, CODE)
#define LABRADOR_LDPC_OUTPUT_LEN(CODE)        LABRADOR_LDPC_PASTE_(LABRADOR_LDPC_OUTPUT_LEN_, CODE)
/* the reference's one-argument helper spellings (labrador_ldpc.h:42, :53, ...), kept for sources that use them */
#define LABRADOR_LDPC_CODE_(CODE)              LABRADOR_LDPC_CODE_##CODE
#define LABRADOR_LDPC_N_(CODE)                 LABRADOR_LDPC_N_##CODE
#define LABRADOR_LDPC_K_(CODE)                 LABRADOR_LDPC_K_##CODE
#define LABRADOR_LDPC_BF_WORKING_LEN_(CODE)    LABRADOR_LDPC_BF_WORKING_LEN_##CODE
#define LABRADOR_LDPC_MS_WORKING_LEN_(CODE)    LABRADOR_LDPC_MS_WORKING_LEN_##CODE
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_(CODE) LABRADOR_LDPC_MS_WORKING_U8_LEN_##CODE
#define LABRADOR_LDPC_OUTPUT_LEN_(CODE)        LABRADOR_LDPC_OUTPUT_LEN_##CODE

/* TC128: n=128 k=64 punctured=0 edges=512 */
#define LABRADOR_LDPC_N_TC128                   (128)
#define LABRADOR_LDPC_K_TC128                   (64)
#define LABRADOR_LDPC_BF_WORKING_LEN_TC128      (128)
#define LABRADOR_LDPC_MS_WORKING_LEN_TC128      (1280)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TC128   (8)
#define LABRADOR_LDPC_OUTPUT_LEN_TC128          (16)

/* TC256: n=256 k=128 punctured=0 edges=1024 */
#define LABRADOR_LDPC_N_TC256                   (256)
#define LABRADOR_LDPC_K_TC256                   (128)
#define LABRADOR_LDPC_BF_WORKING_LEN_TC256      (256)
#define LABRADOR_LDPC_MS_WORKING_LEN_TC256      (2560)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TC256   (16)
#define LABRADOR_LDPC_OUTPUT_LEN_TC256          (32)

/* TC512: n=512 k=256 punctured=0 edges=2048 */
#define LABRADOR_LDPC_N_TC512                   (512)
#define LABRADOR_LDPC_K_TC512                   (256)
#define LABRADOR_LDPC_BF_WORKING_LEN_TC512      (512)
#define LABRADOR_LDPC_MS_WORKING_LEN_TC512      (5120)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TC512   (32)
#define LABRADOR_LDPC_OUTPUT_LEN_TC512          (64)

/* TM1280: n=1280 k=1024 punctured=128 edges=4992 */
#define LABRADOR_LDPC_N_TM1280                  (1280)
#define LABRADOR_LDPC_K_TM1280                  (1024)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM1280     (1408)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM1280     (12160)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM1280  (48)
#define LABRADOR_LDPC_OUTPUT_LEN_TM1280         (176)

/* TM1536: n=1536 k=1024 punctured=256 edges=5888 */
#define LABRADOR_LDPC_N_TM1536                  (1536)
#define LABRADOR_LDPC_K_TM1536                  (1024)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM1536     (1792)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM1536     (15104)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM1536  (96)
#define LABRADOR_LDPC_OUTPUT_LEN_TM1536         (224)

/* TM2048: n=2048 k=1024 punctured=512 edges=7680 */
#define LABRADOR_LDPC_N_TM2048                  (2048)
#define LABRADOR_LDPC_K_TM2048                  (1024)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM2048     (2560)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM2048     (20992)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM2048  (192)
#define LABRADOR_LDPC_OUTPUT_LEN_TM2048         (320)

/* TM5120: n=5120 k=4096 punctured=512 edges=19968 */
#define LABRADOR_LDPC_N_TM5120                  (5120)
#define LABRADOR_LDPC_K_TM5120                  (4096)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM5120     (5632)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM5120     (48640)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM5120  (192)
#define LABRADOR_LDPC_OUTPUT_LEN_TM5120         (704)

/* TM6144: n=6144 k=4096 punctured=1024 edges=23552 */
#define LABRADOR_LDPC_N_TM6144                  (6144)
#define LABRADOR_LDPC_K_TM6144                  (4096)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM6144     (7168)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM6144     (60416)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM6144  (384)
#define LABRADOR_LDPC_OUTPUT_LEN_TM6144         (896)
/* reference spellings of the four entries above (labrador_ldpc.h:76, :88, :100, :112) */
#define LABRADOR_LDPC_BF_WORKING_LEN_TM6140     LABRADOR_LDPC_BF_WORKING_LEN_TM6144
#define LABRADOR_LDPC_MS_WORKING_LEN_TM6140     LABRADOR_LDPC_MS_WORKING_LEN_TM6144
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM6140  LABRADOR_LDPC_MS_WORKING_U8_LEN_TM6144
#define LABRADOR_LDPC_OUTPUT_LEN_TM6140         LABRADOR_LDPC_OUTPUT_LEN_TM6144

/* TM8192: n=8192 k=4096 punctured=2048 edges=30720 */
#define LABRADOR_LDPC_N_TM8192                  (8192)
#define LABRADOR_LDPC_K_TM8192                  (4096)
#define LABRADOR_LDPC_BF_WORKING_LEN_TM8192     (10240)
#define LABRADOR_LDPC_MS_WORKING_LEN_TM8192     (83968)
#define LABRADOR_LDPC_MS_WORKING_U8_LEN_TM8192  (768)
#define LABRADOR_LDPC_OUTPUT_LEN_TM8192         (1280)

/* ======================================================================================
 * Part 1 -- the reference's 21 symbols, same names, signatures and meaning.
 * ====================================================================================== */

/* capi/include/labrador_ldpc.h:118, :121  (capi/src/lib.rs:15-23) */
size_t labrador_ldpc_code_n(enum labrador_ldpc_code code);
size_t labrador_ldpc_code_k(enum labrador_ldpc_code code);

/* capi/include/labrador_ldpc.h:124-135  (capi/src/lib.rs:48-66; src/decoder.rs:93-116) */
size_t labrador_ldpc_bf_working_len(enum labrador_ldpc_code code);
size_t labrador_ldpc_ms_working_u8_len(enum labrador_ldpc_code code);
size_t labrador_ldpc_ms_working_len(enum labrador_ldpc_code code);
size_t labrador_ldpc_output_len(enum labrador_ldpc_code code);

/* capi/include/labrador_ldpc.h:143, :151-152  (capi/src/lib.rs:25-46; src/encoder.rs:293-315).
 * Host-side systematic encoder: first k/8 bytes are data, the rest is written with parity. */
void labrador_ldpc_encode(enum labrador_ldpc_code code, uint8_t *codeword);
void labrador_ldpc_copy_encode(enum labrador_ldpc_code code, const uint8_t *data, uint8_t *codeword);

/* capi/include/labrador_ldpc.h:167-170  (capi/src/lib.rs:68-81; src/decoder.rs:243-301).
 * Bit-flipping decoder (with the erasure pre-pass for punctured codes), one codeword, host
 * pointers, on the GPU.  `input` n/8 bytes, `output` output_len bytes, `working` is accepted and
 * not touched, `iters_run` may be NULL. */
bool labrador_ldpc_decode_bf(enum labrador_ldpc_code code, const uint8_t *input, uint8_t *output,
                             uint8_t *working, size_t max_iters, size_t *iters_run);

/* capi/include/labrador_ldpc.h:193-208  (capi/src/lib.rs:83-127; src/decoder.rs:347-475).
 * One codeword, host pointers; runs the same kernels as the batched calls with batch = 1
 * (every LLR type incl. f64: the register-resident kernels).
 * `llrs` n entries, `output` output_len bytes, `iters_run` may be NULL. */
bool labrador_ldpc_decode_ms_i8 (enum labrador_ldpc_code code, const int8_t  *llrs, uint8_t *output,
                                 int8_t  *working, uint8_t *working_u8, size_t max_iters, size_t *iters_run);
bool labrador_ldpc_decode_ms_i16(enum labrador_ldpc_code code, const int16_t *llrs, uint8_t *output,
                                 int16_t *working, uint8_t *working_u8, size_t max_iters, size_t *iters_run);
bool labrador_ldpc_decode_ms_f32(enum labrador_ldpc_code code, const float   *llrs, uint8_t *output,
                                 float   *working, uint8_t *working_u8, size_t max_iters, size_t *iters_run);
bool labrador_ldpc_decode_ms_f64(enum labrador_ldpc_code code, const double  *llrs, uint8_t *output,
                                 double  *working, uint8_t *working_u8, size_t max_iters, size_t *iters_run);

/* decode_ms::<i32> (src/decoder.rs:60-68): the crate's generic accepts i32 LLRs; the reference's C API does
 * not export it (capi/src/lib.rs:97-127 stops at i8/i16/f32/f64).  Same contract as the four above, with the
 * i32 forms of the LLR helpers below and labrador_ldpc_decode_ms_batch_i32 in part 2. */
bool labrador_ldpc_decode_ms_i32(enum labrador_ldpc_code code, const int32_t *llrs, uint8_t *output,
                                 int32_t *working, uint8_t *working_u8, size_t max_iters, size_t *iters_run);
void labrador_ldpc_hard_to_llrs_i32(enum labrador_ldpc_code code, const uint8_t *input, int32_t *llrs);
void labrador_ldpc_llrs_to_hard_i32(enum labrador_ldpc_code code, const int32_t *llrs, uint8_t *output);

/* capi/include/labrador_ldpc.h:219-226  (capi/src/lib.rs:129-153; src/decoder.rs:484-493) */
void labrador_ldpc_hard_to_llrs_i8 (enum labrador_ldpc_code code, const uint8_t *input, int8_t  *llrs);
void labrador_ldpc_hard_to_llrs_i16(enum labrador_ldpc_code code, const uint8_t *input, int16_t *llrs);
void labrador_ldpc_hard_to_llrs_f32(enum labrador_ldpc_code code, const uint8_t *input, float   *llrs);
void labrador_ldpc_hard_to_llrs_f64(enum labrador_ldpc_code code, const uint8_t *input, double  *llrs);

/* capi/include/labrador_ldpc.h:237-244  (capi/src/lib.rs:155-179; src/decoder.rs:498-509) */
void labrador_ldpc_llrs_to_hard_i8 (enum labrador_ldpc_code code, const int8_t  *llrs, uint8_t *output);
void labrador_ldpc_llrs_to_hard_i16(enum labrador_ldpc_code code, const int16_t *llrs, uint8_t *output);
void labrador_ldpc_llrs_to_hard_f32(enum labrador_ldpc_code code, const float   *llrs, uint8_t *output);
void labrador_ldpc_llrs_to_hard_f64(enum labrador_ldpc_code code, const double  *llrs, uint8_t *output);

/* ======================================================================================
 * Part 2 -- batched GPU entry points (no counterpart in the reference; what a caller that
 * loops over labrador_ldpc_decode_ms_* per frame, e.g. perftest/src/main.rs:9-29, moves to).
 * ====================================================================================== */

/* status codes */
#define LABRADOR_LDPC_HIP_OK            0
#define LABRADOR_LDPC_HIP_EINVAL      (-1)   /* bad code / NULL pointer / misaligned buffer */
#define LABRADOR_LDPC_HIP_ENODEV      (-2)   /* no HIP device, or the device is not gfx950 */
#define LABRADOR_LDPC_HIP_ERUNTIME    (-3)   /* a HIP runtime call failed */
#define LABRADOR_LDPC_HIP_EUNSUPPORTED (-4)  /* valid request this build has no kernel for */

#define LABRADOR_LDPC_HIP_MEM_HOST    0      /* buffers are host memory; the call stages them */
#define LABRADOR_LDPC_HIP_MEM_DEVICE  1      /* buffers are device memory resident on `device` */

#define LABRADOR_LDPC_HIP_DEVICE_CURRENT (-1)  /* the calling thread's current HIP device */
#define LABRADOR_LDPC_HIP_DEVICE_ALL     (-2)  /* MEM_HOST only: shard the batch over every gfx950 device */

/* ABI of the batched entry points.  3 = `struct labrador_ldpc_hip_opts` starts with `struct_size` (this header).
 * (1 = the 24-byte struct of library 0.1.0, 2 = 0.2.0's 32-byte struct with n_devices / devices appended and no way for the
 * library to tell the two apart -- a 0.1.0 client handed 0.2.0 its padding as n_devices.  Both are gone: the shared object
 * carries the soname liblabrador_ldpc_hip.so.3, and labrador_ldpc_hip_abi_version() lets a dlopen() client check.) */
#define LABRADOR_LDPC_HIP_ABI 3

/* Zero-initialise, then set what you need: `struct labrador_ldpc_hip_opts o = {0};` means device 0, host memory, default
 * stream, tuned kernel.  `struct_size` is what makes the struct growable: the library reads a field only if it lies inside
 * the first `struct_size` bytes and takes every field beyond as zero, so a client built against THIS header keeps working
 * with a later library whose struct has grown.  0 (what `= {0}` leaves) stands for this header's layout up to and
 * including `devices`; LABRADOR_LDPC_HIP_OPTS_INIT sets it to the client's own sizeof, which is what a client should use
 * from now on. */
struct labrador_ldpc_hip_opts {
    size_t struct_size; /* sizeof(struct labrador_ldpc_hip_opts) as the CALLER compiled it, or 0 (see above) */
    int   device;     /* HIP device ordinal, LABRADOR_LDPC_HIP_DEVICE_CURRENT or _ALL */
    int   memory;     /* LABRADOR_LDPC_HIP_MEM_HOST or _DEVICE */
    void *stream;     /* hipStream_t to launch on; NULL = the default stream.  With MEM_DEVICE
                         the call only enqueues work and returns (asynchronous); with MEM_HOST
                         it returns after the results are in the host buffers. */
    int   variant;    /* enum labrador_ldpc_hip_variant below; 0 = the tuned default.  Every variant returns identical results. */
    int   n_devices;  /* > 0: shard a MEM_HOST batch over devices[0 .. n_devices) (`device` is ignored) */
    const int *devices; /* HIP ordinals; an ordinal may repeat (that many host pipelines on it, at most four at a time: further
                           repeats queue behind them) */
};
#define LABRADOR_LDPC_HIP_OPTS_INIT { sizeof(struct labrador_ldpc_hip_opts) }

/* `variant`: which of the library's decode_ms kernels a batched call runs.  0 is what callers want; the others exist so that the
 * tuned choice can be A/B-ed against its alternatives (all return identical results; a value that was not built for the code and
 * LLR type yields LABRADOR_LDPC_HIP_EUNSUPPORTED).  One kernel value, optionally OR-ed with flags. */
enum labrador_ldpc_hip_variant {
    LABRADOR_LDPC_HIP_VARIANT_DEFAULT       = 0,    /* the tuned kernel for (code, LLR type, batch size) */
    LABRADOR_LDPC_HIP_VARIANT_IPT1          = 1,    /* f32-pipe kernel (messages as f32 / f64 / i32 registers), one index per thread */
    LABRADOR_LDPC_HIP_VARIANT_IPT2          = 2,    /* ... two indices per thread (t, t + M/2) */
    LABRADOR_LDPC_HIP_VARIANT_IPT4          = 4,    /* ... four */
    LABRADOR_LDPC_HIP_VARIANT_LEAN          = 16,   /* OR-ed with IPTn: the register-lean check phase (one check row at a time) */
    LABRADOR_LDPC_HIP_VARIANT_PAIR          = 32,   /* adjacent-index pair ownership (TM8192, TM2048); f64: OR-ed with IPTn = in-place messages */
    LABRADOR_LDPC_HIP_VARIANT_BITSLICE      = 64,   /* i8 LLRs, TM codes: the bit-sliced kernel whatever the batch size */
    LABRADOR_LDPC_HIP_VARIANT_F64_WORKSPACE = 100,  /* f64: the general kernel with its messages in a device workspace */
    /* flags */
    LABRADOR_LDPC_HIP_VARIANT_STATIC        = 256,  /* fixed-stride distribution of the codewords instead of the launch's queue; with
                                                       BITSLICE: the lockstep kernel instead of the slot-refill one (TM1536, TM1280: by
                                                       name at any batch size, in the default dispatch from 65 536 frames up) */
    LABRADOR_LDPC_HIP_VARIANT_NAN_ONE_PASS  = 512,  /* TM5120 / TM1280 f32: NaN LLRs handled inside the one kernel ... */
    LABRADOR_LDPC_HIP_VARIANT_NAN_TWO_PASS  = 1024  /* ... or by a second launch over marked codewords (the default from ~1000 frames) */
};

/* Multi-GPU (the reference's analogue is perftest/src/main.rs:39-45, one worker per
 * core over independent frames): with MEM_HOST buffers and a device set -- `device` ==
 * LABRADOR_LDPC_HIP_DEVICE_ALL or a `devices` list -- the batched calls split the batch into
 * contiguous slices (labrador_ldpc_hip_shard_range), one per listed device, and run every slice
 * through that device's own copy/kernel/copy pipeline on a worker thread of the library.  No data
 * moves between devices and there is no collective; results land in the caller's buffers exactly
 * as in the single-device call.  `stream` must be NULL.  The first failing slice's status is
 * returned (its text, prefixed with the device, via labrador_ldpc_hip_last_error()). */

/* Decode `batch` independent frames.
 *   llrs    [batch][n]            row-major, n = labrador_ldpc_code_n(code)
 *   output  [batch][output_len]   hard bits incl. punctured parity, MSB first
 *   iters   [batch]               0-based index of the converging iteration, or max_iters
 *   success [batch]               1 if all parity checks were satisfied, else 0
 * Per frame the three results equal what labrador_ldpc_decode_ms_* returns for that frame.
 * `opts` may be NULL (host memory, current device, default stream).  With MEM_DEVICE,
 * `output` must be 8-byte aligned.  Returns a status code. */
int labrador_ldpc_decode_ms_batch_f32(enum labrador_ldpc_code code, const float *llrs, uint8_t *output,
                                      uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_decode_ms_batch_i8 (enum labrador_ldpc_code code, const int8_t *llrs, uint8_t *output,
                                      uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_decode_ms_batch_i16(enum labrador_ldpc_code code, const int16_t *llrs, uint8_t *output,
                                      uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts);
/* i32 (src/decoder.rs:60-68): saturating 32-bit integer arithmetic on the GPU. */
int labrador_ldpc_decode_ms_batch_i32(enum labrador_ldpc_code code, const int32_t *llrs, uint8_t *output,
                                      uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts);
/* f64: same results contract.  By default the register-resident kernels with 64-bit registers and LDS elements (plain for the
 * small codes, register-lean for TM2048 / TM5120, in-place messages for TM6144 / TM8192); `variant`
 * LABRADOR_LDPC_HIP_VARIANT_F64_WORKSPACE (100) names the general fallback that keeps the per-edge messages in a device workspace
 * allocated per call (10-100x slower; csrc/decode_ms_f64.hip). */
int labrador_ldpc_decode_ms_batch_f64(enum labrador_ldpc_code code, const double *llrs, uint8_t *output,
                                      uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts);

/* Device-resident batches on SEVERAL GPUs with one call (the reference's analogue: one job over all workers,
 * perftest/src/main.rs:39-52; capi/src/lib.rs:83-95 for the buffers' meaning).  Part i is frames[i] frames whose four buffers --
 * llrs[i], output[i] (8-byte aligned), iters[i], success[i], laid out as in labrador_ldpc_decode_ms_batch_* -- are DEVICE memory
 * resident on HIP device devices[i]; an ordinal may repeat (several parts on one GPU: up to four run concurrently, more queue
 * behind them) and frames[i] may be 0.  Every part is
 * enqueued by the library's persistent worker thread of its device (pinned to the GPU's NUMA node) on a stream of the library's own
 * and the call returns when ALL parts are decoded; work the caller enqueued on its own streams for these buffers must be complete
 * before the call.  No data crosses between devices and there is no collective.  Returns the first failing part's status
 * (labrador_ldpc_hip_last_error() names the part and its device). */
int labrador_ldpc_decode_ms_batch_f32_multi(enum labrador_ldpc_code code, size_t n_parts, const int *devices, const float *const *llrs,
                                            uint8_t *const *output, uint32_t *const *iters, uint8_t *const *success,
                                            const size_t *frames, size_t max_iters, int variant);
int labrador_ldpc_decode_ms_batch_i8_multi (enum labrador_ldpc_code code, size_t n_parts, const int *devices, const int8_t *const *llrs,
                                            uint8_t *const *output, uint32_t *const *iters, uint8_t *const *success,
                                            const size_t *frames, size_t max_iters, int variant);
int labrador_ldpc_decode_ms_batch_i16_multi(enum labrador_ldpc_code code, size_t n_parts, const int *devices, const int16_t *const *llrs,
                                            uint8_t *const *output, uint32_t *const *iters, uint8_t *const *success,
                                            const size_t *frames, size_t max_iters, int variant);
int labrador_ldpc_decode_ms_batch_i32_multi(enum labrador_ldpc_code code, size_t n_parts, const int *devices, const int32_t *const *llrs,
                                            uint8_t *const *output, uint32_t *const *iters, uint8_t *const *success,
                                            const size_t *frames, size_t max_iters, int variant);
int labrador_ldpc_decode_ms_batch_f64_multi(enum labrador_ldpc_code code, size_t n_parts, const int *devices, const double *const *llrs,
                                            uint8_t *const *output, uint32_t *const *iters, uint8_t *const *success,
                                            const size_t *frames, size_t max_iters, int variant);

/* Batched bit-flipping decoder (src/decoder.rs:243-301), the batched form of
 * labrador_ldpc_decode_bf:  input [batch][n/8], output [batch][output_len], iters [batch]
 * (bit-flipping iterations + erasure iterations, or that sum's maximum on failure), success [batch]. */
int labrador_ldpc_decode_bf_batch(enum labrador_ldpc_code code, const uint8_t *input, uint8_t *output,
                                  uint32_t *iters, uint8_t *success, size_t batch, size_t max_iters,
                                  const struct labrador_ldpc_hip_opts *opts);

/* Batched systematic encoder on the GPU: codewords[f] = copy_encode(data[f]) for every frame
 * (src/encoder.rs:293-315; the per-frame C entry is labrador_ldpc_copy_encode above).
 *   data      [batch][k/8]   MSB-first bytes
 *   codewords [batch][n/8]   first k/8 bytes = data, rest = parity
 * Host or device buffers per opts->memory (device buffers 4-byte aligned); asynchronous with
 * MEM_DEVICE.  Returns a status code. */
int labrador_ldpc_encode_batch(enum labrador_ldpc_code code, const uint8_t *data, uint8_t *codewords,
                               size_t batch, const struct labrador_ldpc_hip_opts *opts);

/* Batched LLR helpers: the data formats either side of decode_ms for whole batches -- hard_to_llrs
 * (src/decoder.rs:484-493; per-frame C entries capi/src/lib.rs:129-153) and llrs_to_hard (src/decoder.rs:498-509;
 * capi/src/lib.rs:155-179), frame after frame:
 *   input / output  [batch][n/8]  packed bits, MSB first
 *   llrs            [batch][n]    -1 for a set bit, +1 for a clear one; a bit is set where the LLR is < 0
 * With opts->memory == MEM_DEVICE the conversion is a streaming kernel on opts->stream (asynchronous; llrs
 * 16-byte aligned), so that hard decisions produced on the device (encode_batch, decode_bf_batch,
 * a decode_ms_batch output) feed decode_ms_batch without leaving HBM.  With host buffers (opts NULL or MEM_HOST) the
 * frames are converted in place by the library's host code -- the data is there and the loop is cheaper than the PCIe
 * crossing; opts->device / devices are ignored.  Returns a status code. */
int labrador_ldpc_hard_to_llrs_batch_i8 (enum labrador_ldpc_code code, const uint8_t *input, int8_t  *llrs, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hard_to_llrs_batch_i16(enum labrador_ldpc_code code, const uint8_t *input, int16_t *llrs, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hard_to_llrs_batch_i32(enum labrador_ldpc_code code, const uint8_t *input, int32_t *llrs, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hard_to_llrs_batch_f32(enum labrador_ldpc_code code, const uint8_t *input, float   *llrs, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hard_to_llrs_batch_f64(enum labrador_ldpc_code code, const uint8_t *input, double  *llrs, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_llrs_to_hard_batch_i8 (enum labrador_ldpc_code code, const int8_t  *llrs, uint8_t *output, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_llrs_to_hard_batch_i16(enum labrador_ldpc_code code, const int16_t *llrs, uint8_t *output, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_llrs_to_hard_batch_i32(enum labrador_ldpc_code code, const int32_t *llrs, uint8_t *output, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_llrs_to_hard_batch_f32(enum labrador_ldpc_code code, const float   *llrs, uint8_t *output, size_t batch, const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_llrs_to_hard_batch_f64(enum labrador_ldpc_code code, const double  *llrs, uint8_t *output, size_t batch, const struct labrador_ldpc_hip_opts *opts);

/* Synthetic AWGN frames on the device (harness side of the path; what perftest's ms_trial does
 * per frame at perftest/src/main.rs:10-18, batched): frame f takes codeword (f mod pool) of
 * `codewords` ([pool][n/8] bytes, MSB first), maps bit b to 1-2b, adds sigma*N(0,1) from a
 * counter-based generator keyed by (seed, f, sample), and writes
 *   f32: the sample itself;   i8: clamp(round(scale*sample), -lim, lim).
 * All pointers are DEVICE memory (opts->memory is ignored); asynchronous on opts->stream. */
int labrador_ldpc_hip_awgn_f32(enum labrador_ldpc_code code, const uint8_t *codewords, size_t pool,
                               float *llrs, size_t batch, float sigma, uint64_t seed,
                               const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hip_awgn_i8 (enum labrador_ldpc_code code, const uint8_t *codewords, size_t pool,
                               int8_t *llrs, size_t batch, float sigma, float scale, int lim,
                               uint64_t seed, const struct labrador_ldpc_hip_opts *opts);
/* The same for frames [first_frame, first_frame + batch) of a larger job: frame f of the call is global frame
 * first_frame + f -- it takes codeword ((first_frame + f) mod pool) and the generator stream of that global index --
 * so the shards of a job generated on N devices are, byte for byte, the slices of the buffer one call with
 * first_frame = 0 writes (labrador_ldpc_hip_awgn_f32 / _i8 are these with first_frame = 0).  This is what makes an
 * N-GPU run of the harness decode the very frames of the one-GPU run (perftest/src/main.rs:39-52: one job, N workers). */
int labrador_ldpc_hip_awgn_f32_at(enum labrador_ldpc_code code, const uint8_t *codewords, size_t pool,
                                  float *llrs, uint64_t first_frame, size_t batch, float sigma, uint64_t seed,
                                  const struct labrador_ldpc_hip_opts *opts);
int labrador_ldpc_hip_awgn_i8_at (enum labrador_ldpc_code code, const uint8_t *codewords, size_t pool,
                                  int8_t *llrs, uint64_t first_frame, size_t batch, float sigma, float scale, int lim,
                                  uint64_t seed, const struct labrador_ldpc_hip_opts *opts);

/* Harness diagnostic: the shader clock (MHz) `device` (an ordinal or LABRADOR_LDPC_HIP_DEVICE_CURRENT) holds under a full-chip
 * VALU load lasting `busy_ms` milliseconds (0.01 .. 1000): the advance of the shader-clock counter against the 100 MHz
 * real-time counter, median over the workgroups.  Synchronous, default stream.  A multi-GPU harness prints it per worker
 * beside the worker's rate (perftest/src/main.rs:39-52 aggregates workers it assumes equal; GPUs of one node are not).
 * Returns a status code. */
int labrador_ldpc_hip_shader_clock_mhz(int device, double busy_ms, double *mhz);

/* Edge stream CRC of this library's own code tables, computed like the reference's
 * test_iter_parity (src/codes/mod.rs:508-533).  Lets a test pin the tables the kernels are
 * generated from against the reference's nine known answers without a GPU. */
uint32_t labrador_ldpc_hip_edge_crc(enum labrador_ldpc_code code);

/* The parity-check edges (check, variable) of this library's code tables in the order of the reference's
 * LDPCCode::iter_paritychecks() (src/codes/mod.rs:435-441, body :275-362; variables n .. n+p-1 are the
 * punctured ones).  Writes up to `cap` pairs (either array may be NULL) and returns the number of edges
 * (= paritycheck_sum, src/codes/mod.rs:405-409); 0 for a bad code. */
size_t labrador_ldpc_hip_edges(enum labrador_ldpc_code code, uint16_t *checks, uint16_t *variables, size_t cap);

/* The contiguous slice [*first, *first + *count) of `batch` frames that part `index` of `parts`
 * takes in a sharded call (slices differ by at most one frame).  Returns a status code. */
int labrador_ldpc_hip_shard_range(size_t batch, size_t parts, size_t index, size_t *first, size_t *count);

/* Number of HIP devices usable by this library (gfx950 only); 0 if none. Never fails. */
int labrador_ldpc_hip_device_count(void);

/* Human-readable description of the calling thread's last failure ("" if none).  The reference-shaped single-frame calls
 * (labrador_ldpc_decode_ms_*, labrador_ldpc_decode_bf) keep the reference's signature and can only return `false` when the
 * library could not run at all (no GPU, a HIP failure, a bad code): they then zero `output`, set *iters_run = max_iters, and
 * leave the reason here -- check it to tell "did not converge" from "did not run" (with LABRADOR_LDPC_HIP_VERBOSE=1 in the
 * environment the reason is also written to stderr, once per distinct failure site). */
const char *labrador_ldpc_hip_last_error(void);

/* Library version string. */
const char *labrador_ldpc_hip_version(void);

/* Identity of the loaded library's BUILD: 16 hex digits, a hash of what determines the code object -- the sources of
 * labrador_ldpc_amd/csrc, this header, the compiler flags and the compiler's version (csrc/build_id.sh) -- not of the produced
 * bytes, which hipcc does not reproduce bit for bit.  Two builds of one source tree report one id; any source or flag edit
 * changes it.  Profiles record the id they were collected on (profiles/hbm_traffic.json, bench.py). */
const char *labrador_ldpc_hip_build_id(void);

/* Name of the kernel labrador_ldpc_decode_ms_batch_i8 launches for a 4-byte-aligned device batch of `batch` frames with this
 * `variant` ("decode_ms_bs_kernel" / "decode_ms_bs_refill_kernel" / "decode_ms_bs_split_kernel" / "decode_ms_bs_split_refill_kernel": bit-sliced; "decode_ms_pair_kernel" /
 * "decode_ms_kernel": the f32-pipe kernels) -- the default dispatch depends on the batch size; harnesses label their
 * measurements with it.  The launcher and this function share one predicate (csrc/decode_ms_i8.hip: pick_i8_kernel).
 * "" for a bad code and for a request this build has no kernel for (the batched call then returns LABRADOR_LDPC_HIP_EUNSUPPORTED);
 * buffers that are NOT 4-byte aligned never take the bit-sliced kernels (default dispatch: the f32-pipe kernel of the code;
 * `variant` 64: EUNSUPPORTED). */
const char *labrador_ldpc_hip_decode_ms_i8_kernel(enum labrador_ldpc_code code, int variant, size_t batch);

/* The LABRADOR_LDPC_HIP_ABI the loaded library was built with: a client that dlopen()s the library compares it with its
 * own header's before passing a struct labrador_ldpc_hip_opts. */
int labrador_ldpc_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LABRADOR_LDPC_HIP_H */
