/* ldpc_decode.c -- instantiates decode_ms<T> and the LLR helpers for the five types the
 * reference implements `DecodeFrom` for (src/decoder.rs:42-86).
 *
 * TEST INFRASTRUCTURE (see ldpc_oracle.h).  Build with -fno-fast-math -ffp-contract=off:
 * float add/sub must be single IEEE operations, as in the Rust original.
 */
#include "ldpc_oracle.h"
#include "ldpc_internal.h"

#include <limits.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* Integer DecodeFrom: saturating_abs / saturating_add / saturating_sub,
 * src/decoder.rs:42-68.  Done in a wider type then clamped. */
static inline int64_t clamp64(int64_t x, int64_t lo, int64_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ---- i8: src/decoder.rs:42-50 ---- */
#define T int8_t
#define SUF i8
#define T_MAX INT8_MAX
#define T_ABS(x)    ((int8_t)clamp64((x) < 0 ? -(int64_t)(x) : (int64_t)(x), INT8_MIN, INT8_MAX))
#define T_ADD(a, b) ((int8_t)clamp64((int64_t)(a) + (int64_t)(b), INT8_MIN, INT8_MAX))
#define T_SUB(a, b) ((int8_t)clamp64((int64_t)(a) - (int64_t)(b), INT8_MIN, INT8_MAX))
#include "ldpc_decode_tmpl.h"
#undef T
#undef SUF
#undef T_MAX
#undef T_ABS
#undef T_ADD
#undef T_SUB

/* ---- i16: src/decoder.rs:51-59 ---- */
#define T int16_t
#define SUF i16
#define T_MAX INT16_MAX
#define T_ABS(x)    ((int16_t)clamp64((x) < 0 ? -(int64_t)(x) : (int64_t)(x), INT16_MIN, INT16_MAX))
#define T_ADD(a, b) ((int16_t)clamp64((int64_t)(a) + (int64_t)(b), INT16_MIN, INT16_MAX))
#define T_SUB(a, b) ((int16_t)clamp64((int64_t)(a) - (int64_t)(b), INT16_MIN, INT16_MAX))
#include "ldpc_decode_tmpl.h"
#undef T
#undef SUF
#undef T_MAX
#undef T_ABS
#undef T_ADD
#undef T_SUB

/* ---- i32: src/decoder.rs:60-68 ---- */
#define T int32_t
#define SUF i32
#define T_MAX INT32_MAX
#define T_ABS(x)    ((int32_t)clamp64((x) < 0 ? -(int64_t)(x) : (int64_t)(x), INT32_MIN, INT32_MAX))
#define T_ADD(a, b) ((int32_t)clamp64((int64_t)(a) + (int64_t)(b), INT32_MIN, INT32_MAX))
#define T_SUB(a, b) ((int32_t)clamp64((int64_t)(a) - (int64_t)(b), INT32_MIN, INT32_MAX))
#include "ldpc_decode_tmpl.h"
#undef T
#undef SUF
#undef T_MAX
#undef T_ABS
#undef T_ADD
#undef T_SUB

/* Float DecodeFrom: abs clears the sign bit (src/decoder.rs:73, :82); add/sub are the plain
 * IEEE operators (:74-75, :83-84). */
static inline float abs_bits_f32(float x)
{
    uint32_t b; memcpy(&b, &x, 4); b &= 0x7FFFFFFFu; memcpy(&x, &b, 4); return x;
}
static inline double abs_bits_f64(double x)
{
    uint64_t b; memcpy(&b, &x, 8); b &= 0x7FFFFFFFFFFFFFFFull; memcpy(&x, &b, 8); return x;
}

/* ---- f32: src/decoder.rs:69-77 ---- */
#define T float
#define SUF f32
#define T_MAX FLT_MAX
#define T_ABS(x)    abs_bits_f32(x)
#define T_ADD(a, b) ((a) + (b))
#define T_SUB(a, b) ((a) - (b))
#include "ldpc_decode_tmpl.h"
#undef T
#undef SUF
#undef T_MAX
#undef T_ABS
#undef T_ADD
#undef T_SUB

/* ---- f64: src/decoder.rs:78-86 ---- */
#define T double
#define SUF f64
#define T_MAX DBL_MAX
#define T_ABS(x)    abs_bits_f64(x)
#define T_ADD(a, b) ((a) + (b))
#define T_SUB(a, b) ((a) - (b))
#include "ldpc_decode_tmpl.h"
#undef T
#undef SUF
#undef T_MAX
#undef T_ABS
#undef T_ADD
#undef T_SUB
