// decode_ms_kernel.hpp -- the batched min-sum decoder kernel for gfx950 (MI355X).
//
// Replaces LDPCCode::decode_ms<T> (/root/reference/src/decoder.rs:347-475) and the edge
// iterator it is driven by (/root/reference/src/codes/mod.rs:275-362) for batches of
// independent codewords.  Results (hard bits, iterations, success) are bit-identical to
// the reference's; the formulation is not a translation of it:
//
//  * INDEX-ALIGNED OWNERSHIP.  Every code is built from MxM blocks that are either a
//    shifted identity or one of the CCSDS permutations pi_k.  A thread owns an index i
//    (IPT of them) and with it check i of every block row and variable i of every block
//    column.  An unshifted identity block then connects a check and a variable of the SAME
//    thread: its messages never leave registers (7 of the 15 blocks of TM2048/TM8192).
//    Only shifted/permuted blocks exchange data, through two LDS arrays, and because lanes
//    run along i and every block is a rotation (inside quarters, for pi_k) each LDS access
//    of a wave is unit-stride with at most one wrap: conflict-free, no index tables.
//
//  * MESSAGE STATE IN REGISTERS.  The reference keeps u[E], v[E], min1[C], min2[C] and a
//    sign bitmap in memory (decoder.rs:375-378) and derives u from them at the start of the
//    next iteration (decoder.rs:391-405).  Here the check side derives the next u at the END
//    of its update, as "smallest |v| among the OTHER edges of the check" with the product of
//    the other edges' signs -- the same value decoder.rs:391-405 selects through min1/min2
//    (proof in DESIGN.md) -- so per edge only u and v are kept, in VGPRs, for the whole
//    decode.  LLRs are read from HBM once, hard bits written once.
//
//  * ORDER.  Marginals are accumulated per variable in the reference's edge order
//    restricted to that variable (LLR first, then blocks by (block row, term)), with the
//    same single IEEE / saturating operations (decoder.rs:408), so floating-point and
//    saturating-integer results agree bit for bit.  Min, sign and parity accumulation are
//    order-independent (decoder.rs:430-447).
//
// One iteration = variable phase (marginals, decoder.rs:382-411) | barrier | check phase
// (new v with self-correction, next u, parity; decoder.rs:419-450) | barrier.  The "all
// parities satisfied" vote (decoder.rs:453) is an LDS flag read after the barrier.
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <type_traits>

#include "codes.hpp"
#include "notify.hpp"

#define LDPC_INLINE __attribute__((always_inline))

#include "decode_ms_tuning.hpp"    // tuned settings (kernel experiments and timing diagnostics: tools/kbench/)

// Workgroup barrier for LDS hand-offs.  Written out (instead of __syncthreads()) so that it waits
// for LDS operations only and not for the asynchronous LLR staging copies counted in vmcnt.
#define LDPC_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#if LDPC_PRIO == 1
#define LDPC_SETPRIO(n) do { if constexpr (PRIO_WAVES) { __builtin_amdgcn_s_setprio(n); __builtin_amdgcn_sched_barrier(0); } } while (0)
#elif LDPC_PRIO == 2
#define LDPC_SETPRIO(n) do { if constexpr (PRIO_WAVES) __builtin_amdgcn_s_setprio(n); } while (0)
#else
#define LDPC_SETPRIO(n) do { } while (0)
#endif
#define LDPC_DEV __device__ __forceinline__

namespace ldpc {

// ---- compile-time loop ------------------------------------------------------------------
template <int N> struct IC { static constexpr int value = N; constexpr operator int() const { return N; } };
template <int B, int E, class F>
LDPC_DEV void static_for(F &&f)
{
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

// XOR of D words with three-input XORs (v_bitop3_b32 issues at the fast VALU rate on gfx950)
template <int D>
LDPC_DEV int xor_reduce(const int (&w)[D])
{
    int acc = w[0];
    static_for<0, (D - 1) / 2>([&](auto I_) LDPC_INLINE {
        constexpr int i = 1 + 2 * decltype(I_)::value;
        acc = __builtin_amdgcn_bitop3_b32(acc, w[i], w[i + 1], 0x96);
    });
    if constexpr (D % 2 == 0) acc ^= w[D - 1];
    return acc;
}

// acc ^ w[0] ^ ... ^ w[N - 1], two words per v_bitop3_b32 (a running `acc ^= w` costs one v_xor per word)
template <int N, int D>
LDPC_DEV int xor_into(int acc, const int (&w)[D])
{
    static_assert(N <= D);
    static_for<0, N / 2>([&](auto I_) LDPC_INLINE {
        constexpr int i = 2 * decltype(I_)::value;
        acc = __builtin_amdgcn_bitop3_b32(acc, w[i], w[i + 1], 0x96);
    });
    if constexpr (N % 2 == 1) acc ^= w[N - 1];
    return acc;
}

// ---- prototype analysis -------------------------------------------------------------------
constexpr bool blk_local(const Block &b) { return b.kind == BLK_I && b.val == 0; }
constexpr int count_exchanged(const Prototype &p)
{
    int c = 0;
    for (int b = 0; b < p.n_blocks; ++b) c += blk_local(p.blk[b]) ? 0 : 1;
    return c;
}
// slot of block b among the exchanged (non-local) blocks, -1 if local
constexpr int exch_slot(const Prototype &p, int b)
{
    if (blk_local(p.blk[b])) return -1;
    int c = 0;
    for (int i = 0; i < b; ++i) c += blk_local(p.blk[i]) ? 0 : 1;
    return c;
}
constexpr bool col_exchanged(const Prototype &p, int col)
{
    for (int b = 0; b < p.n_blocks; ++b)
        if (p.blk[b].col == col && !blk_local(p.blk[b])) return true;
    return false;
}
constexpr int count_exch_cols(const Prototype &p)
{
    int c = 0;
    for (int col = 0; col < p.n_cols; ++col) c += col_exchanged(p, col) ? 1 : 0;
    return c;
}
// slot of block column `col` among the columns whose marginals are exchanged, -1 if none
constexpr int col_slot(const Prototype &p, int col)
{
    if (!col_exchanged(p, col)) return -1;
    int c = 0;
    for (int i = 0; i < col; ++i) c += col_exchanged(p, i) ? 1 : 0;
    return c;
}
constexpr int row_degree(const Prototype &p, int row)
{
    int c = 0;
    for (int b = 0; b < p.n_blocks; ++b) c += p.blk[b].row == row ? 1 : 0;
    return c;
}
// block index of the j-th block of block row `row`
constexpr int row_block(const Prototype &p, int row, int j)
{
    for (int b = 0; b < p.n_blocks; ++b)
        if (p.blk[b].row == row && j-- == 0) return b;
    return -1;
}

// ---- arithmetic per LLR type: DecodeFrom, decoder.rs:22-86 ----------------------------------
// Register values are kept so that "negative" (hard_bit, decoder.rs:49/:76) is exactly bit 31
// of the 32-bit pattern: true for two's-complement ints, and true for floats because no
// value in this kernel is ever -0.0 (LLRs are canonicalised with +0.0 on load; sums and
// differences of such values cannot produce -0.0; see DESIGN.md "signed zeros").
template <class T> struct Ops;

template <> struct Ops<float> {                       // decoder.rs:69-77
    using R = float;                                   // register type
    using E = float;                                   // LDS exchange element type
    LDPC_DEV static R zero() { return 0.0f; }
    LDPC_DEV static R maxval() { return FLT_MAX; }                              // :72
    // -0.0 -> +0.0, and every NaN -> +inf.
    // NaN LLRs: the reference's hard_bit is `x < 0.0` (:76), false for a NaN whatever its sign bit, while this kernel reads
    // "negative" from bit 31, so a sign-carrying NaN must not reach the registers.  Clearing that bit on the common path
    // (compare + select, integer bit tricks, an asm bundle, a ballot and a cold fix-up branch: all tried) cost the kernels at
    // their register limit 20-50 spilled registers.  But a NaN LLR and a +inf LLR are THE SAME INPUT to decode_ms -- every
    // output bit, the iteration count and the success flag agree:
    //   * the marginal is NaN + u resp. inf + u, for ever (u is finite: +-min1 / min2 <= maxval, :391-405): never `< 0`, so
    //     hard bit 0 (:457, :469) and no contribution to the parity (:445);
    //   * every message along the variable's edges is NaN - u resp. inf - u = the same again; it is kept by the
    //     self-correction (old v is 0, then itself: `hard_bit() ==` holds, :422), is never `< 0` (no sign contribution, :439),
    //     and its magnitude NaN resp. inf passes neither `< min1` nor `< min2` (:430, :433) nor `== min1` (:391): the checks
    //     see an edge that takes no part in the minima, and its own u is min1 either way.
    // So the load maps NaN to +inf with one v_min_f32 -- minNum(NaN, inf) = inf; the add before it quiets a signalling NaN,
    // which minNum would otherwise turn into a quiet NaN.  +inf LLRs were always part of the contract (tests since round 1).
    // (Written with the builtin, not as inline asm: an asm statement between a load and its use makes the compiler wait
    // for every LLR load on the spot -- ten serialised L2 round trips per variable phase in the register-lean kernel.)
    LDPC_DEV static R load(float x) { return __builtin_fminf(x + 0.0f, __builtin_inff()); }
    // LATE canonicalisation, for the kernels where the load-time form does not come for free.  Without the NaN mapping the
    // compiler never materialised `llr = raw + 0.0`: it kept the raw registers and folded the `+ 0.0` into the copy that starts
    // each marginal's accumulation.  A two-operation load() cannot be folded that way; it becomes a second set of values, and
    // the two kernels that sit at a forced register limit (TM1280 f32 at 168, the register-lean TM5120 f32 at 128) spilled
    // 20-45 more registers for it (-17 % / -16 %, profiles/r03_kbench/kb4.txt).  Those kernels keep the RAW LLR (keep_raw), add
    // it as it is, and canonicalise the finished marginal instead: raw + u1 + ... equals llr + u1 + ... as a VALUE at every
    // step (a -0.0 addend behaves like +0.0 unless everything is -0.0), `+ 0.0` then turns a -0.0 result into +0.0, and a NaN
    // LLR leaves a NaN marginal, which min(., inf) maps to the +inf marginal a +inf LLR would have left.  One more v_min per
    // transmitted column and iteration, no extra registers.
    LDPC_DEV static R canon_late(R x) { return __builtin_fminf(x + 0.0f, __builtin_inff()); }
    LDPC_DEV static R keep_raw(float x) { return x; }
    LDPC_DEV static R load_nonan(float x) { return x + 0.0f; }                   // for values a vote has shown to hold no NaN
    LDPC_DEV static float store(R x) { return x; }
    LDPC_DEV static R from_lds(float x) { return x; }                           // already canonical
    LDPC_DEV static int bits(R x) { return __float_as_int(x); }
    static constexpr bool SIGN_WORD_IS_BIT31_ONLY = true;      // (the register-lean check phases form it from raw bits themselves)
    LDPC_DEV static int sign_word(R x) { return __float_as_int(x) & (int)0x80000000; }     // bit 31 of a message, alone
    LDPC_DEV static R add(R a, R b) { return a + b; }                           // :74
    LDPC_DEV static R sub(R a, R b) { return a - b; }                           // :75
    LDPC_DEV static R sub_nv(R a, R b) { return a - b; }                        // the new v of an edge (:421)
    LDPC_DEV static R mag(R x) { return __builtin_fabsf(x); }                   // :73 (may be +inf)
    // min of magnitudes.  AX/AY/AZ say whether the operand is a signed message whose magnitude is
    // meant (the |x| source modifier is free) or already a magnitude.  Written as asm so that the
    // operation tree of exclusive_min() is emitted as designed: through fminf() LLVM re-associates
    // it into ~50 % more v_min ops plus canonicalising v_max ops.
    template <bool AX, bool AY>
    LDPC_DEV static R min2(R x, R y)
    {
        R d;
        if constexpr (AX && AY) asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(x), "v"(y));
        else if constexpr (AX) asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(d) : "v"(x), "v"(y));
        else if constexpr (AY) asm("v_min_f32_e64 %0, %1, |%2|" : "=v"(d) : "v"(x), "v"(y));
        else asm("v_min_f32_e32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
        return d;
    }
    // min(maxval, ...): the cap comes from an SGPR
    template <bool AX>
    LDPC_DEV static R min2_cap(R x)
    {
        R d;
        const float cap = FLT_MAX;
        if constexpr (AX) asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(d) : "v"(x), "s"(cap));
        else asm("v_min_f32_e32 %0, %2, %1" : "=v"(d) : "v"(x), "s"(cap));
        return d;
    }
    template <bool AX, bool AY>
    LDPC_DEV static R min3_cap(R x, R y)
    {
        R d;
        const float cap = FLT_MAX;
        if constexpr (AX && AY) asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(d) : "v"(x), "v"(y), "s"(cap));
        else if constexpr (!AX && !AY) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(cap));
        else d = min2_cap<false>(min2<AX, AY>(x, y));
        return d;
    }
    template <bool AX, bool AY, bool AZ>
    LDPC_DEV static R min3(R x, R y, R z)
    {
        R d;
        if constexpr (AX && AY && AZ) asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(x), "v"(y), "v"(z));
        else if constexpr (AX && AY) asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
        else if constexpr (!AX && !AY && !AZ) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
        else d = min2<false, AZ>(min2<AX, AY>(x, y), z);
        return d;
    }
    // magnitude `m` (>= 0) with the sign taken from bit 31 of `s`
    LDPC_DEV static R with_sign(R m, int s)
    {
        return __int_as_float((__float_as_int(m) & 0x7FFFFFFF) | (s & (int)0x80000000));
    }
    LDPC_DEV static R select_zero(bool z, R x) { return z ? 0.0f : x; }
    // x where p is not a negative non-zero number, else +0: "bits(p) <= 0x80000000" as the borrow of an integer
    // subtraction (an I-class VOP2 operation that pairs with the F and I classes) instead of a float compare (C class)
    LDPC_DEV static R keep_unless_negative(R p, R x)
    {
        // (the same through __builtin_usub_overflow, which lets the compiler place the wait states, measures within
        // +-1 % of this bundle on every kernel)
        R r;
        asm("v_subrev_co_u32_e32 %0, vcc, %2, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, 0, %3, vcc"
            : "=&v"(r) : "v"(p), "s"(0x80000001u), "v"(x) : "vcc");
        return r;
    }
    // Self-correction test of decoder.rs:422: drop nv iff old != 0 and sign(nv) != sign(old).
    // `old` with its sign flipped when nv is negative is a negative NON-ZERO float exactly then
    // (old is never -0.0), so one three-input bit op (old ^ (nv & 0x80000000)) and one float
    // compare decide it; "-0.0 < 0" is false, which is the old == 0 case.  (A NaN nv -- always the positive quiet
    // one, see load() -- leaves t = old: dropped iff old < 0, kept otherwise, as `NaN.hard_bit() == old.hard_bit()`.)
    LDPC_DEV static bool drop(R nv, R old)
    {
        const int t = __builtin_amdgcn_bitop3_b32(__float_as_int(old), __float_as_int(nv), (int)0x80000000, 0x78);
        return __int_as_float(t) < 0.0f;
    }
    // Self-correction as a CLAMP (FORM 2 / 3 / 5).  "Keep nv iff it lies on old's side of zero (any side if old == 0)" is
    // v = median(nv, 0, X) for any X with X = nv when old == 0 and, when old != 0, the sign of old and |X| >= |nv|.
    //   FORM 2:  X = fma(old, big, nv)   (one v_fma_f32: full rate on gfx950, tools/ubench/valu_rate.hip) -- needs
    //            big * |old| > |nv| for every nonzero old and every nv of the decode, which the caller guarantees (integer
    //            messages: big = 2^20; f32: the tightened range vote, nocap_limit_for());
    //   FORM 3:  X = nv + mul_legacy(old, inf): +-inf for every nonzero old (denormals included), 0 * inf = 0 under
    //            the legacy rule, so no range condition beyond "nothing is NaN or infinite";
    //   FORM 5:  the same decision without a median: three full-rate operations, none of the 4-cycle class.
    // One v_med3_f32 replaces the compare and the select of forms 0 / 1, and with them the VCC round trip between the two
    // (two wait states on gfx950): sub, fma, med3 instead of sub, mul, cmp, cndmask.  Per edge update 0.94 ns against
    // 1.13-1.25 ns per instruction slot in the micro-benchmark (profiles/r03_final/valu_rate.txt).
    template <int FORM>
    LDPC_DEV static R clamp_to_side(R nv, R old, float big)
    {
        static_assert(FORM == 2 || FORM == 3 || FORM == 5, "clamp forms of the self-correction: 2, 3, 5 (6: integer messages only, IntOps)");
        R x, r;
        if constexpr (FORM == 2) {
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "s"(big), "v"(old), "v"(nv));
        } else if constexpr (FORM == 5) {
            // No C-class instruction at all: s = clamp01(1 + nv * old * 2^100) is 1 unless the signs differ (then nv * old
            // < 0, at least 2^-86 in magnitude for values >= 2^-43, and 1 - 2^14 clamps to 0); v = nv * s + 0 (the + 0 keeps
            // a dropped negative nv from becoming -0.0).  Needs only "no product underflows", like the multiply form.
            float pr, sel;
            asm("v_mul_f32_e32 %0, %1, %2" : "=v"(pr) : "v"(nv), "v"(old));
            asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(sel) : "v"(pr), "s"(0x1p100f));
            asm("v_fma_f32 %0, %1, %2, 0" : "=v"(r) : "v"(nv), "v"(sel));
            return r;
        } else {
            asm("v_mul_legacy_f32_e64 %0, %1, %2" : "=v"(x) : "v"(old), "s"(__builtin_inff()));
            x = x + nv;
        }
        asm("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(nv), "v"(x));
        return r;
    }
    // nv, or +0 where drop(nv, old)  (zeroing by EXEC predication instead of v_cndmask measured slower:
    // EXEC writes stall the VALU -- DESIGN.md 4.4)
    // FORM: 0 = compare + select, 1 = the select through keep_unless_negative (chosen per kernel, see selfcorr_form())
    // (the clamp forms need a guarantee about the values: self_correct_b)
    template <int FORM>
    LDPC_DEV static R self_correct(R nv, R old)
    {
        if constexpr (FORM == 1) {
            const int t = __builtin_amdgcn_bitop3_b32(__float_as_int(old), __float_as_int(nv), (int)0x80000000, 0x78);
            return keep_unless_negative(__int_as_float(t), nv);
        } else {
            return select_zero(drop(nv, old), nv);
        }
    }
    // The same for codewords whose LLRs passed the range vote (BOUNDED: every |LLR| <= nocap_limit and every
    // nonzero |LLR| >= 2^-20, see begin_codeword): "old != 0 and the signs differ" is then exactly "nv * old < 0".
    // No value is infinite (the nocap bound), and every value of the decode is a multiple of g = 2^(e_min - 23),
    // e_min >= -20 the exponent of the smallest nonzero |LLR| (sums and differences of multiples of g round to
    // multiples of g), so a nonzero value is at least 2^-43 and a product of two cannot underflow; nv == 0 gives
    // v = 0 whichever way the test goes.  An F-class v_mul_f32 in place of the VOP3 bit operation: +1 %.
    // FORM 2 / 3: the clamp forms above (FORM 2 with big = 2^126: the launch's limit then also keeps
    // 2^126 * 2^-43 above every magnitude of the decode, nocap_limit_for()).
    template <bool BOUNDED, int FORM = 0>
    LDPC_DEV static R self_correct_b(R nv, R old)
    {
        if constexpr (BOUNDED && FORM >= 2) {
            return clamp_to_side<FORM>(nv, old, 0x1p126f);
        } else if constexpr (BOUNDED) {
            float p;
            asm("v_mul_f32_e32 %0, %1, %2" : "=v"(p) : "v"(nv), "v"(old));
            if constexpr (FORM == 1) return keep_unless_negative(p, nv);
            else return select_zero(p < 0.0f, nv);
        } else {
            return self_correct<(FORM == 1 ? 1 : 0)>(nv, old);
        }
    }
    // m >= 0 has bit 31 clear, so "m with sign s_all ^ s_own" is one three-input XOR of sign words
    LDPC_DEV static R apply_sign(R m, int s_all, int s_own)
    {
        return __int_as_float(__builtin_amdgcn_bitop3_b32(__float_as_int(m), s_all, s_own, 0x96));
    }
};

// f64 LLRs (decoder.rs:78-86): two VGPRs per value, the sign is bit 31 of the HIGH word, so the
// sign-word machinery (bits / apply_sign) works on that word; no -0.0 either (load adds +0.0).
// Plain expressions instead of pinned instruction trees: f64 is the least used variant and its
// VALU ops are quarter rate whatever the tree looks like.
template <> struct Ops<double> {
    using R = double;
    using E = double;
    LDPC_DEV static R zero() { return 0.0; }
    LDPC_DEV static R maxval() { return DBL_MAX; }
    LDPC_DEV static R load(double x) { return __builtin_fmin(x + 0.0, __builtin_inf()); }      // -0.0 -> +0.0, NaN -> +inf: see Ops<float>::load
    LDPC_DEV static R canon_late(R x) { return __builtin_fmin(x + 0.0, __builtin_inf()); }
    LDPC_DEV static R keep_raw(double x) { return x; }
    LDPC_DEV static R load_nonan(double x) { return x + 0.0; }
    LDPC_DEV static double store(R x) { return x; }
    LDPC_DEV static R from_lds(double x) { return x; }
    LDPC_DEV static int bits(R x) { return __double2hiint(x); }
    static constexpr bool SIGN_WORD_IS_BIT31_ONLY = true;
    LDPC_DEV static int sign_word(R x) { return __double2hiint(x) & (int)0x80000000; }
    LDPC_DEV static R add(R a, R b) { return a + b; }
    LDPC_DEV static R sub(R a, R b) { return a - b; }
    LDPC_DEV static R sub_nv(R a, R b) { return a - b; }
    LDPC_DEV static R mag(R x) { return __builtin_fabs(x); }
    template <bool AX, bool AY>
    LDPC_DEV static R min2(R x, R y)
    {
        const R a = AX ? __builtin_fabs(x) : x, b = AY ? __builtin_fabs(y) : y;
        return __builtin_fmin(a, b);              // v_min_f64: a NaN operand is ignored, as the reference's `<` does (:430-434)
    }
    template <bool AX> LDPC_DEV static R min2_cap(R x) { return min2<AX, false>(x, DBL_MAX); }
    template <bool AX, bool AY> LDPC_DEV static R min3_cap(R x, R y) { return min2<false, false>(min2<AX, AY>(x, y), DBL_MAX); }
    template <bool AX, bool AY, bool AZ>
    LDPC_DEV static R min3(R x, R y, R z) { return min2<false, AZ>(min2<AX, AY>(x, y), z); }
    template <int FORM>
    LDPC_DEV static R self_correct(R nv, R old)                                  // decoder.rs:422-425
    {
        return (old != 0.0 && (nv < 0.0) != (old < 0.0)) ? 0.0 : nv;
    }
    template <bool BOUNDED, int FORM = 0> LDPC_DEV static R self_correct_b(R nv, R old) { return self_correct<0>(nv, old); }
    LDPC_DEV static R apply_sign(R m, int s_all, int s_own)
    {
        return __hiloint2double(__double2hiint(m) ^ s_all ^ s_own, __double2loint(m));
    }
};

// Integer LLR types run on the float pipeline: i8/i16 values and every intermediate of the
// algorithm are integers of magnitude <= 2^16, which f32 represents exactly, so saturating
// add/sub (decoder.rs:47-48, :56-57) are an f32 add/sub followed by a clamp, and all the sign-bit
// and exclusive-minimum machinery of the f32 path applies unchanged.  saturating_abs(-2^(b-1)) =
// 2^(b-1)-1 (decoder.rs:46, :55) falls out of capping the exclusive minimum at maxval.
template <class I, int LO, int HI> struct IntOps : Ops<float> {     // decoder.rs:42-59
    LDPC_DEV static R maxval() { return (float)HI; }
    LDPC_DEV static R load(I x) { return (float)(int)x; }                       // never -0.0
    LDPC_DEV static R canon_late(R x) { return x; }
    LDPC_DEV static R keep_raw(I x) { return (float)(int)x; }
    LDPC_DEV static R load_nonan(I x) { return (float)(int)x; }
    LDPC_DEV static R clamp(R x) { return __builtin_amdgcn_fmed3f(x, (float)LO, (float)HI); }
    LDPC_DEV static R add(R a, R b) { return clamp(a + b); }                    // saturating_add
    LDPC_DEV static R sub(R a, R b) { return clamp(a - b); }                    // saturating_sub
    // The new v of an edge, decoder.rs:421, WITHOUT the clamp of saturating_sub: v is only ever used through its
    // sign, its zero-ness (both unchanged by the clamp) and min(|v|, maxval) inside the capped exclusive minimum
    // (saturating_abs of the clamped value IS min(|a - b|, maxval), whichever end clamped), so the clamp is dead
    // work; a - b is exact in f32 (|a - b| < 2^17).  One v_med3 less per edge and iteration.
    LDPC_DEV static R sub_nv(R a, R b) { return a - b; }
    LDPC_DEV static R mag(R x) { return __builtin_fminf(__builtin_fabsf(x), (float)HI); }   // saturating_abs
    // (the sign word of an integer message as 0.0 * x = +-0.0 -- a float multiply instead of a v_and -- measures 0 to -2 %:
    // profiles/r03_kbench/kb19_sign_by_mul.txt)
    // Self-correction test of decoder.rs:422 for integer-valued messages: old != 0 and the signs differ exactly
    // when the product is negative -- |nv|, |old| < 2^17, so the f32 product can neither underflow to zero nor
    // lose its sign (it may round), and nv == 0 gives v = 0 whichever way the test goes.  An F-class v_mul_f32
    // (co-issues with the 4-cycle instructions of other waves) instead of the VOP3 bit operation of the f32 path,
    // where the same trick would need a vote on the LLR range (+1 %).
    LDPC_DEV static bool drop(R nv, R old)
    {
        float p;
        asm("v_mul_f32_e32 %0, %1, %2" : "=v"(p) : "v"(nv), "v"(old));
        return p < 0.0f;
    }
    // FORM 2 / 3: the clamp forms of Ops<float>::clamp_to_side -- exact for every integer message: |nv| < 2^17 and a
    // nonzero |old| >= 1, so 2^20 * |old| > |nv| always
    // FORM 6: form 5 without its multiply.  Integer messages are zero or at least 1 in magnitude, so nv * old is <= -1 when
    // the signs differ, >= 1 when they agree and 0 when either is zero: s = clamp01(fma(nv, old, 1)) is 0 / 1 / 1 with no scale
    // factor, and v = fma(nv, s, 0).  Two full-rate instructions per update and no 4-cycle one (form 2: fma + med3, form 5:
    // mul + fma + fma).  The product is below 2^33 and rounds, but never across zero.
    template <int FORM>
    LDPC_DEV static R self_correct(R nv, R old)
    {
        if constexpr (FORM == 6) {
            float sel, r;
            asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(sel) : "v"(nv), "v"(old));
            asm("v_fma_f32 %0, %1, %2, 0" : "=v"(r) : "v"(nv), "v"(sel));
            return r;
        } else if constexpr (FORM >= 2) {
            return Ops<float>::clamp_to_side<FORM>(nv, old, 0x1p20f);
        } else if constexpr (FORM == 1) {
            float p;
            asm("v_mul_f32_e32 %0, %1, %2" : "=v"(p) : "v"(nv), "v"(old));
            return Ops<float>::keep_unless_negative(p, nv);
        } else {
            return Ops<float>::select_zero(drop(nv, old), nv);
        }
    }
    template <bool BOUNDED, int FORM = 0> LDPC_DEV static R self_correct_b(R nv, R old) { return self_correct<FORM>(nv, old); }
    template <bool AX>
    LDPC_DEV static R min2_cap(R x)
    {
        R d;
        const float cap = (float)HI;
        if constexpr (AX) asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(d) : "v"(x), "s"(cap));
        else asm("v_min_f32_e32 %0, %2, %1" : "=v"(d) : "v"(x), "s"(cap));
        return d;
    }
    template <bool AX, bool AY>
    LDPC_DEV static R min3_cap(R x, R y)
    {
        R d;
        const float cap = (float)HI;
        if constexpr (AX && AY) asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(d) : "v"(x), "v"(y), "s"(cap));
        else if constexpr (!AX && !AY) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(cap));
        else d = min2_cap<false>(Ops<float>::min2<AX, AY>(x, y));
        return d;
    }
};
template <> struct Ops<int8_t>  : IntOps<int8_t, -128, 127> {};
template <> struct Ops<int16_t> : IntOps<int16_t, -32768, 32767> {};

// i32 LLRs (decoder.rs:60-68): genuine 32-bit integer arithmetic -- the f32 pipeline is exact only to 2^24.
// Saturating add/sub are v_add_i32 / v_sub_i32 with the clamp bit.  Those, every integer min / max / compare and the three-operand
// integer forms issue at HALF the rate of v_xor / v_sub_u32 / v_ashrrev / v_bitop3 on gfx950 (tools/ubench/wide_rate.hip: 1.8 against
// 0.95 ns per wave-instruction and SIMD), which is why decode_ms::<i32> runs at about half decode_ms::<f32>'s rate (f32 add / sub / fma
// are full rate, |x| is a free source modifier there and the sign application one v_bitop3).  Round 6 moved what it could to
// full-rate operations (TM8192 i32 4.24 -> see profiles/r06_final/rates_all_codes.txt):
//   * the self-correction without a compare (self_correct below: four full-rate operations for xor + two v_cmp + s_and + select);
//   * the MAGNITUDE as the wrapping |x| (one v_sub_u32 + one v_max_i32 instead of v_sub_i32 clamp + v_max_i32).  It wraps where
//     saturating_abs saturates -- |INT_MIN| comes out as 0x80000000 instead of INT_MAX (:64) -- so magnitudes are compared UNSIGNED and
//     every exclusive minimum is capped at maxval = INT_MAX (which decoder.rs:414-415 does anyway: min1 / min2 start there):
//     min(sat|a|, ...) = min_u32(wrap|a|, INT_MAX, ...).  The cap rides in a v_min3_u32's third operand except on rows of degree 2 and
//     4-6 (one operation more per three edges).
//   (Sign words as x >> 31 -- which would also save the shift in apply_sign -- were measured at the compiler: 118 spilled registers
//   in the TM8192 pair kernel against 1; not adopted.)
// "Negative" is bit 31, so the sign-word machinery applies unchanged.  The LDS element is a 4-byte container (float) holding the
// integer's bits, which lets the pair kernel's 64-bit accesses carry it.
template <> struct Ops<int32_t> {
    using R = int;
    using E = float;
    LDPC_DEV static R zero() { return 0; }
    LDPC_DEV static R maxval() { return 0x7FFFFFFF; }                           // :63
    LDPC_DEV static R load(int32_t x) { return x; }
    LDPC_DEV static R canon_late(R x) { return x; }
    LDPC_DEV static R keep_raw(int32_t x) { return x; }
    LDPC_DEV static R load_nonan(int32_t x) { return x; }
    LDPC_DEV static float store(R x) { return __int_as_float(x); }
    LDPC_DEV static R from_lds(float x) { return __float_as_int(x); }
    LDPC_DEV static int bits(R x) { return x; }
    static constexpr bool SIGN_WORD_IS_BIT31_ONLY = true;
    LDPC_DEV static int sign_word(R x) { return x & (int)0x80000000; }
    LDPC_DEV static R add(R a, R b) { R d; asm("v_add_i32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }   // :65
    LDPC_DEV static R sub(R a, R b) { R d; asm("v_sub_i32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }   // :66
    LDPC_DEV static R sub_nv(R a, R b) { return sub(a, b); }                    // (32-bit: the clamp is what keeps it from wrapping)
    // |x| as an UNSIGNED word (wraps INT_MIN to 0x80000000 where :64 saturates to INT_MAX): see above
    LDPC_DEV static R mag(R x) { const R s = x >> 31; return (R)((unsigned)(x ^ s) - (unsigned)s); }
    LDPC_DEV static R umin(R a, R b) { return (unsigned)b < (unsigned)a ? b : a; }
    template <bool AX, bool AY>
    LDPC_DEV static R min2(R x, R y)
    {
        const R a = AX ? mag(x) : x, b = AY ? mag(y) : y;
        return umin(a, b);
    }
    template <bool AX> LDPC_DEV static R min2_cap(R x) { return umin(AX ? mag(x) : x, maxval()); }
    template <bool AX, bool AY> LDPC_DEV static R min3_cap(R x, R y) { return umin(min2<AX, AY>(x, y), maxval()); }
    template <bool AX, bool AY, bool AZ>
    LDPC_DEV static R min3(R x, R y, R z) { return min2<false, AZ>(min2<AX, AY>(x, y), z); }
    // decoder.rs:422-425: v = 0 where the old v is non-zero and of the other sign, else nv -- without a compare: bit 31 of
    // (nv ^ old) & (old | -old) says "drop" (old | -old has bit 31 set exactly for old != 0, INT_MIN included); an arithmetic shift
    // spreads it over the word and it is cleared out of nv: v_sub_u32, v_bitop3, v_ashrrev, v_bitop3 -- all full rate
    template <int FORM>
    LDPC_DEV static R self_correct(R nv, R old)
    {
        const int negold = (int)(0u - (unsigned)old);
        const int drop = __builtin_amdgcn_bitop3_b32(nv, old, negold, 0x2C) >> 31;          // (a ^ b) & (b | c)
        return __builtin_amdgcn_bitop3_b32(nv, drop, drop, 0x30);                           // a & ~b
    }
    template <bool BOUNDED, int FORM = 0> LDPC_DEV static R self_correct_b(R nv, R old) { return self_correct<0>(nv, old); }
    // m >= 0 negated when the product of the other edges' signs is negative (:398-405)
    LDPC_DEV static R apply_sign(R m, int s_all, int s_own)
    {
        const int k = (s_all ^ s_own) >> 31;                                     // 0 or -1
        return (m ^ k) - k;
    }
};

// e[i] = min(maxval, min over j != i of a[j]).  Equals what decoder.rs:391-395 selects from
// (min1, min2): min2 if |v_i| is a smallest magnitude of the check, min1 otherwise -- and
// min1/min2 start at maxval (decoder.rs:414-415) and are only replaced by strictly smaller
// values (:430-434), hence the clamp.  Elements are grouped in threes so that one min3 per
// element finishes the job: ~1.7 operations per edge at degree 6, ~1.9 at degree 18.
template <class O, int D, bool ABS, bool CAP = true>
LDPC_DEV void exclusive_min(const typename O::R (&a)[D], typename O::R (&e)[D])
{
    // CAP = false: the caller guarantees that every magnitude is below maxval (then the clamp is the
    // identity); only checks of degree >= 4 save operations by it
    using R = typename O::R;
    const R MX = O::maxval();
    if constexpr (D == 1) {
        e[0] = MX;
    } else if constexpr (D == 2) {
        if constexpr (CAP) {
            e[0] = O::template min2_cap<ABS>(a[1]);
            e[1] = O::template min2_cap<ABS>(a[0]);
        } else {
            e[0] = ABS ? O::mag(a[1]) : a[1];
            e[1] = ABS ? O::mag(a[0]) : a[0];
        }
    } else if constexpr (D == 3) {
        e[0] = O::template min3_cap<ABS, ABS>(a[1], a[2]);
        e[1] = O::template min3_cap<ABS, ABS>(a[0], a[2]);
        e[2] = O::template min3_cap<ABS, ABS>(a[0], a[1]);
    } else {
        constexpr int G = (D + 2) / 3;
        R t[G], x[G];
        static_for<0, G>([&](auto g_) LDPC_INLINE {
            constexpr int g = decltype(g_)::value, n = (3 * g + 3 <= D) ? 3 : D - 3 * g;
            if constexpr (n == 3) t[g] = O::template min3<ABS, ABS, ABS>(a[3 * g], a[3 * g + 1], a[3 * g + 2]);
            else if constexpr (n == 2) t[g] = O::template min2<ABS, ABS>(a[3 * g], a[3 * g + 1]);
            else t[g] = ABS ? O::mag(a[3 * g]) : a[3 * g];
        });
        exclusive_min<O, G, false, CAP>(t, x);
        static_for<0, G>([&](auto g_) LDPC_INLINE {
            constexpr int g = decltype(g_)::value, n = (3 * g + 3 <= D) ? 3 : D - 3 * g;
            if constexpr (n == 3) {
                e[3 * g]     = O::template min3<ABS, ABS, false>(a[3 * g + 1], a[3 * g + 2], x[g]);
                e[3 * g + 1] = O::template min3<ABS, ABS, false>(a[3 * g],     a[3 * g + 2], x[g]);
                e[3 * g + 2] = O::template min3<ABS, ABS, false>(a[3 * g],     a[3 * g + 1], x[g]);
            } else if constexpr (n == 2) {
                e[3 * g]     = O::template min2<ABS, false>(a[3 * g + 1], x[g]);
                e[3 * g + 1] = O::template min2<ABS, false>(a[3 * g],     x[g]);
            } else {
                e[3 * g] = x[g];
            }
        });
    }
}

// ---- LDS layout of one codeword: [ first half of the xu slots | xva columns | rest of xu | flags ]
constexpr int lds_xu_off(const Prototype &p, int slot, int blk_bytes)
{
    const int lo = (count_exchanged(p) + 1) / 2;
    return (slot < lo ? slot : slot + count_exch_cols(p)) * blk_bytes;
}
constexpr int lds_xva_off(const Prototype &p, int cs, int blk_bytes)
{
    return ((count_exchanged(p) + 1) / 2 + cs) * blk_bytes;
}
// lower of the two region offsets an exchanged block b touches (folded into its address register)
constexpr int lds_bias(const Prototype &p, int b, int blk_bytes)
{
    const int a0 = lds_xu_off(p, exch_slot(p, b), blk_bytes);
    const int a1 = lds_xva_off(p, col_slot(p, p.blk[b].col), blk_bytes);
    return a0 < a1 ? a0 : a1;
}

// rank of local edge (S, B) among a thread's local edges, index-major
constexpr int local_edge_rank(const Prototype &p, int S, int B)
{
    int nloc = 0, r = 0;
    for (int b = 0; b < p.n_blocks; ++b)
        if (blk_local(p.blk[b])) { if (b < B) ++r; ++nloc; }
    return S * nloc + r;
}
// local-edge updates done in the variable phase (see LOCAL_IN_VAR in the kernel body)
template <int CODE, class T, int IPT, int LEAN>
constexpr int local_in_var_default()
{
    // pays where ONE workgroup has a CU to itself (nothing else fills the idle VALU of its variable phase):
    // TM6144 f32 0/3/5/9/12 -> 10.83 / 11.05 / 11.12 / 11.13 / 11.14 M codewords/s; no effect on TM2048,
    // TM1536, TM1280, where several workgroups share a CU
    // (i8/i16 TM6144: 10.08 / 9.91 / 9.86 at 0 / 4 / 9 -- their variable phase carries the clamps already)
    return (CODE == TM6144 && IPT == 1 && LEAN == 0 && std::is_same_v<T, float>) ? 9 : 0;
}

// Self-correction select through an integer borrow (Ops<float>::keep_unless_negative) instead of a float compare:
// 0 = no, 1 = the i8/i16 types, 2 = f32 as well.  Pays on the rate-4/5 codes, whose degree-18 checks make the check
// phase the most C-class heavy (TM5120 i8 17.4 -> 18.0 at 4 dB, 7.03 -> 7.31 at 2 dB, f32 17.4 -> 18.4; TM1280 i8
// 63.8 -> 65.2, f32 66.4 -> 67.9; TM1536 +1 % M codewords/s); neutral or negative elsewhere (TM2048 -1..-3 %,
// TM6144 -2.5 %, TM8192 pair kernel -1.5..-3.5 %: there the compiler fills the two wait states between the VCC write
// and the select with other work, which it cannot do inside the asm bundle).
template <int CODE>
constexpr int selfcorr_carry_default()
{
    return (CODE == TM1280 || CODE == TM1536 || CODE == TM5120) ? 2 : 0;
}

template <int CODE, int IPT>
constexpr int Geometry_G() { return CODES[CODE].m / IPT >= 64 ? 1 : 64 / (CODES[CODE].m / IPT); }      // codewords per workgroup (Geometry::G)

// Form of the self-correction in decode_ms_kernel (Ops<float>::clamp_to_side): 0 = compare / borrow + select, 2 = v_fma +
// v_med3, 3 = v_mul_legacy + v_add + v_med3, 5 = three full-rate operations and no median, 6 = two (integer messages only:
// IntOps::self_correct).  This is the DEFAULT of the
// kernel's FORM template parameter; for the f32 kernels with a clamp-free loop the launcher instantiates 2 and 3 and picks by
// max_iters (form 2 narrows the range vote: nocap_limit_for()).
// Same-process A/B (tools/kbench.hip, profiles/r03_kbench/kb11_forms.txt; identical outputs), M codewords/s, forms 0 / 2 / 3 / 5:
//   TM6144 i8 11.63 / 12.26 / - / 12.13;  TM2048 i8 46.9 / 49.37 / 47.74 / 48.26;  TC512 i8 (3 dB) 312.5 / 318.4 / - / -;
//   the register-lean kernels (form 0 = the borrow form there): TM5120 i8 4 dB 20.76 / 20.74 / - / 21.08, 2 dB 7.87 / 7.85 / - /
//   7.98; TM1280 i8 77.96 / - / - / 81.44 -- their chunked check rows like the all-full-rate form best;
//   f32 (only inside a clamp-free copy of the loop): TM2048 - / 40.29 / 38.23 / 38.19;  TC512 - / 139.96 / 136.63 / 136.80.
// Form 6 against the best of those on every i8 kernel (profiles/r03_kbench/kb18_form6.txt; identical outputs): TM5120 4 dB
// 21.90 -> 22.61, 2 dB 8.29 -> 8.58; TM8192 pair kernel 7.78 -> 7.99; TM2048 49.4 -> 51.4; TM1536 63.1 -> 65.9; TM1280 85.6 -> 87.4;
// TM6144 12.28 -> 12.44; TC512 683 -> 701, TC256 1 128 -> 1 142, TC128 2 155 -> 2 190 (5 dB): the narrow types' default everywhere.
template <int CODE, class T>
constexpr int selfcorr_med3()
{
    if (sizeof(T) > 4 || std::is_same_v<T, int32_t>) return 0;
    if (LDPC_SELFCORR_MED3 >= 0) return LDPC_SELFCORR_MED3;
    constexpr bool narrow = sizeof(T) <= 2;
    if (narrow) return 6;
    return (CODE == TM2048 || CODE == TC512 || CODE == TM1536) ? 2 : 0;    // f32: the kernels with a clamp-free loop (has_nocap_loop)
}

// f32 kernels that carry a second, clamp-free copy of their iteration loop (NOCAP_POSSIBLE in the kernel body)
template <int CODE, class T, int IPT, int LEAN>
constexpr bool has_nocap_loop()
{
    // (TM1536: 65.7 -> 69.2 M codewords/s with the loop and the clamp form; TM6144 loses 2 % -- its local-edge updates
    // in the variable phase keep the multiply / clamp forms out; TM1280 spills: profiles/r03_kbench/kb14.txt)
    constexpr bool code = CODE == TM2048 || CODE == TC512 || CODE == TM1536;
    return LDPC_NOCAP != 0 && std::is_same_v<T, float> && code && Geometry_G<CODE, IPT>() == 1 && LEAN == 0 && IPT == 1;
}

// The in-phase verdict of the register-lean kernels (LEAN_VERDICT in the kernel body).  Measured and refuted in round 3
// (profiles/r03_kbench/kb3.txt): carrying the rows' state across the vote costs 67 spilled registers at the lean kernels'
// 128-register budget -- TM5120 i8 19.04 -> 17.66 (4 dB), 7.33 -> 6.71 (2 dB), f32 13.7 -> 9.6 M codewords/s.  Off.
template <int CODE, class T>
constexpr bool lean_verdict_default() { return false; }

// Packed LLR registers in the register-lean kernels of the narrow types (PACKED_LLR in the kernel body)
template <int CODE, class T>
constexpr bool packed_llr_default() { return true; }

// Kernels that run iteration 0 as a pass of its own (PEEL_FIRST in the kernel body).  Same-process A/B, M codewords/s:
//   TC128 f32 1863 -> 2067, i8 1806 -> 2061; TC256 894 -> 1025 / 946 -> 1095; TC512 539 -> 598 / 555 -> 676 (5 dB);
//   +3-7 % at 3 dB; config 2 (TC512 f32, 2 dB) 104.8 -> 109.2;
//   TM2048 f32 43.1 -> 44.8, i8 42.6 -> 44.7, config 3 (2 dB) 34.6 -> 35.7; TM1536 f32 62.6 -> 62.9, i8 57.7 -> 59.2;
//   TM6144 i8 10.79 -> 11.10; TM1280 i8 65.1 -> 68.1 -- but TM1280 f32 70.3 -> 63.5 (23 spilled registers at its 168).
// The pair kernel has its own switch (LDPC_PAIR_PEEL_FIRST: TM8192 f32 7.51 -> 7.70).
template <int CODE, class T, int IPT>
constexpr bool peel_first_default()
{
    if (CODE <= TC512) return !std::is_same_v<T, double> || CODE == TC512;      // (f64: TC128 703 -> 483, TC256 384 -> 267, TC512 206 -> 250)
    if (IPT != 1) return false;
    // every type (i32 +2-6 %, f64 0-6 %); TM5120 = the lean kernel: i8 17.9 -> 18.4 at 4 dB; TM6144 f32 10.69 -> 11.42
    if (CODE == TM2048 || CODE == TM1536 || CODE == TM5120 || CODE == TM6144) return true;
    if (CODE == TM1280) return !std::is_same_v<T, float>;
    return false;
}

// What the first of two NaN passes leaves in iters_out for a codeword with a NaN LLR (decode_ms_body, NANPASS): no iteration
// count -- the launcher runs two passes only for max_iters below it
constexpr uint32_t NAN_MARK = 0xFFFFFFFFu;

// Most codeword groups a workgroup takes per draw from the launch's queue (decode_ms_body, "dynamic distribution").  The
// draws of a launch are atomics on ONE address, and the device sustains 85-90 million of those per second whatever else it
// does (measured: TC512 at 5 dB through the queue with two groups per draw stops at 172 M codewords/s against 600 with the
// fixed stride, TC128 with four codewords per group at 672 against 1 900; profiles/r03_kbench/rates_all_queue_everywhere.txt).
// So the queue is for the codes whose decode takes tens of microseconds, with enough groups per draw to stay an order of
// magnitude under that ceiling even at high SNR; the launcher (launch_cfg) takes fewer per draw when a launch is short.
template <int CODE, class T, int IPT>
constexpr uint32_t claim_chunk()
{
    return CODE == TM2048 ? 4 : (CODE == TM5120 || CODE == TM6144) ? 2 : 1;
}

// ---- kernel geometry -----------------------------------------------------------------------
template <int CODE, class T, int IPT>
struct Geometry {
    static constexpr int M = CODES[CODE].m;
    static constexpr int NT = M / IPT;                       // threads per codeword
    static constexpr int G = NT >= 64 ? 1 : 64 / NT;         // codewords per workgroup
    static constexpr int WG = NT * G;                        // workgroup size
    static constexpr int NB = CODES[CODE].proto->n_blocks;
    static constexpr int NROWS = CODES[CODE].proto->n_rows;
    static constexpr int NCOLS = CODES[CODE].proto->n_cols;
    static constexpr int NTX = CODES[CODE].n / M;            // transmitted block columns
    static constexpr int NX = count_exchanged(*CODES[CODE].proto);
    static constexpr int NXC = count_exch_cols(*CODES[CODE].proto);
    static constexpr int OUT_LEN = CODES[CODE].output_len();
    static constexpr size_t LDS_BYTES = (size_t)G * (NX + NXC) * M * sizeof(typename Ops<T>::E) + G * 2 * sizeof(int);
    static_assert(M % IPT == 0 && NT >= 8 && (NT & (NT - 1)) == 0, "bad IPT");
};

// pi_k(i) for check index i whose quarter j = i / (M/4) the caller supplies: a literal when a
// thread's indices never leave a quarter, a wave-uniform scalar when waves do not straddle
// quarters (then the selects below are scalar), a per-lane value otherwise.
template <int K, int M>
LDPC_DEV int pi_dev(int i, int j)
{
    constexpr int LQ = ilog2(M / 4), Q = M / 4;
    constexpr int P0 = phi_of(K, 0, M), P1 = phi_of(K, 1, M), P2 = phi_of(K, 2, M), P3 = phi_of(K, 3, M);
    constexpr int TH = theta_of(K);
    int phi;
    if constexpr (Q <= 256) {
        // the four rotations of a block (each < Q <= 256) packed into one literal and picked by a bit-field
        // extract: one VALU operation.  Written as a chain of selects on the per-lane quarter the compiler
        // emitted EXEC-masked branches, one pair per quarter and edge (119 of them in the TM1280 kernel).
        constexpr unsigned PACK = (unsigned)P0 | ((unsigned)P1 << 8) | ((unsigned)P2 << 16) | ((unsigned)P3 << 24);
        phi = (int)__builtin_amdgcn_ubfe(PACK, (unsigned)j * 8u, 8u);
    } else {
        // rotations up to 16 bits: two literals, one select on bit 1 of the quarter, one bit-field extract
        constexpr unsigned LO = (unsigned)P0 | ((unsigned)P1 << 16), HI = (unsigned)P2 | ((unsigned)P3 << 16);
        const unsigned w = (j & 2) ? HI : LO;
        phi = (int)__builtin_amdgcn_ubfe(w, ((unsigned)j & 1u) * 16u, 16u);
    }
    return (((TH + j) & 3) << LQ) + ((phi + i) & (Q - 1));
}

// JW >= 0: the body is specialised for waves whose indices start in quarter JW (the kernel
// branches once, wave-uniformly, into the matching copy) so that every rotation constant of the
// pi_k blocks is a literal; JW < 0: generic body, constants in SGPRs.
// NANPASS: how NaN LLRs are handled (Ops<float>::load) -- 0 = in line, by this kernel alone; 1 / 2 = the two passes of the
// kernel that cannot afford that (the register-lean f32 one: two_pass_nan() in decode_ms_launch.hpp): pass 1 decodes as if no
// LLR were a NaN and MARKS the codewords that have one (iters_out = NAN_MARK), pass 2 is the in-line kernel over the marked
// codewords only.
// Which kernels keep their channel LLRs in LDS instead of registers (one LDS read per variable phase and transmitted block column;
// the codeword's own lane reads what it wrote: no synchronisation).  TC512 f32: its eight LLR registers are the difference between
// three and four waves per SIMD (min_waves_per_simd()); 2 KB more LDS per wave, 16 waves x 10 000 bytes = a CU's 160 KB.
// TC128 f32 likewise (148 registers; at 128 it keeps 14 values in scratch around the loops, none inside: +10 % at 2 and 3 dB, +-0 at
// 5 dB).  TC256 f32 (154 registers, 19 spilled at 128) gains 8 % / 4 % at 2 / 3 dB and LOSES 13 % at 5 dB, where a decode is two
// iterations and the spills around them weigh: it stays at three waves (profiles/r06_kbench/tc512_llr_lds.txt).  The narrow types
// pack their LLRs already.
template <int CODE, class T, int IPT, int LEAN>
constexpr bool llr_in_lds() { return (CODE == TC512 || CODE == TC128) && std::is_same_v<T, float> && IPT == 1 && LEAN == 0; }

template <int CODE, class T, int IPT, bool PF, int LEAN, int JW, int FORM, int NANPASS = 0>
LDPC_DEV void decode_ms_body(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                             uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                             uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim, uint32_t claim_k, char *lds, char *stage)
{
    using GEO = Geometry<CODE, T, IPT>;
    using O = Ops<T>;
    using R = typename O::R;
    constexpr Prototype P = *CODES[CODE].proto;
    constexpr int M = GEO::M, NT = GEO::NT, G = GEO::G, NB = GEO::NB, NROWS = GEO::NROWS,
                  NCOLS = GEO::NCOLS, NTX = GEO::NTX, NX = GEO::NX, NXC = GEO::NXC;
    constexpr int N = CODES[CODE].n;
    constexpr bool PRIO_WAVES = NT >= 512;       // see LDPC_PRIO
    (void)PRIO_WAVES;

    // LDS, per codeword of the workgroup: [ xu: NX blocks | xva: NXC block columns | 2 flags ]
    //   xu   check -> variable messages of the exchanged blocks, stored at the VARIABLE's index
    //   xva  marginals of the block columns those blocks touch
    //   flag "some parity check failed", double-buffered over iterations
    using E = typename O::E;
    constexpr int SZ = sizeof(E);             // LDS element size (HBM elements are sizeof(T))
    // The xva regions sit in the middle of the xu slots so that, for every exchanged edge, its xu
    // slot and its xva column are less than 64 KB apart: one address register (biased by the
    // lower of the two region offsets, lds_bias()) then serves both accesses through the 16-bit
    // instruction offset.
    // In-place mode (LEAN == 2): [ xu: NX blocks | hi: NXC block columns of 4-byte sign words | flags ].
    // The variable thread overwrites each exchanged u with nv = va - u in the slot it read it from,
    // and publishes the high word (sign) of each exchanged marginal; no array of marginals.
    constexpr bool INPLACE = LEAN == 2;
    constexpr int BLK_BYTES = M * SZ;
    constexpr int FLAG_OFF = INPLACE ? NX * BLK_BYTES + NXC * M * 4 : (NX + NXC) * M * SZ;
    auto hi_off = [](int cs) constexpr { return NX * BLK_BYTES + cs * M * 4; };
    (void)hi_off;
    // LLRs in LDS instead of registers (llr_in_lds()): a region of NTX blocks behind the flag words
    constexpr bool LLR_LDS = llr_in_lds<CODE, T, IPT, LEAN>();
    constexpr int LLR_OFF = FLAG_OFF + 16;
    constexpr int GROUP_BYTES = (FLAG_OFF + 8 + 15) / 16 * 16 + (LLR_LDS ? NTX * M * SZ : 0);
    static_assert(!LLR_LDS || (!INPLACE && !PF && FLAG_OFF % 16 == 0));
    static_assert(!PF || (G == 1 && NT >= 64), "LLR staging needs whole waves per codeword");
    constexpr int TSZ = sizeof(T);
    constexpr int STAGE_BYTES = PF ? N * TSZ : 0;
    (void)STAGE_BYTES;

    const int tid = threadIdx.x;
    const int grp = G == 1 ? 0 : tid / NT;
    // in a quarter-specialised body (JW >= 0) the thread index can be written with its quarter as a literal,
    // which lets the compiler fold the high bits of every address: +2.7 % on TM6144 and TM2048 (and on the
    // TM8192 pair kernel), but -6 % on the lean TM5120 kernel and -2 % on the two-index TM8192 one
    // (register allocation), hence the condition
    constexpr bool LITERAL_QUARTER_T = JW >= 0 && LEAN == 0 && IPT == 1;
    const int t = LITERAL_QUARTER_T ? (tid & (M / 4 - 1)) + JW * (M / 4) : (G == 1 ? tid : tid % NT);
    __builtin_assume(t >= 0 && t < NT);
    // Persistent workgroups over the codeword groups, in chunks of CLAIM_K groups: workgroup b starts with chunk b and
    // then takes chunks b + gridDim.x, ... (claim == nullptr) or whichever chunk is next in the launch's queue (below)
    const uint32_t n_groups = (batch + G - 1) / G;
    uint32_t cw = blockIdx.x * G + grp;
    bool live = cw < batch;
    char *const gbase = lds + (G == 1 ? 0 : grp * GROUP_BYTES);
    auto lds_at = [&](int byte_off) LDPC_INLINE -> E & { return *reinterpret_cast<E *>(gbase + byte_off); };
    auto lds_load = [&](int byte_off) LDPC_INLINE -> E { return lds_at(byte_off); };
    auto lds_store = [&](int byte_off, E val) LDPC_INLINE { lds_at(byte_off) = val; };
    auto flag_at = [&](uint32_t which) LDPC_INLINE -> int & {
        return *reinterpret_cast<int *>(gbase + FLAG_OFF + 4 * (which & 1));
    };
    // f32, one codeword per workgroup: the clamp of the exclusive minimum at FLT_MAX (decoder.rs:414-415)
    // can only bite if some magnitude reaches FLT_MAX, i.e. if an LLR is infinite or so large that sums
    // overflow.  With every |LLR| <= nocap_limit (a bound the host derives from max_iters, see
    // nocap_limit_for() in decode_ms_launch.hpp) nothing can, and the check phase runs without the clamp
    // operations (TM8192 pair kernel +3 %).  The vote is one LDS word per codeword.
    // Measured: TM2048 41.3 -> 43.0 M codewords/s; TM6144 -1.6 %, TM1536 -0.7 %, TM1280 -11 %, and 34 spilled VGPRs
    // with two indices per thread (the second copy of the loop is not free); TC512: 19 spilled VGPRs at the four-waves budget,
    // 108 -> 100 M codewords/s on config 2.  Hence TM2048 only.
    // (TC512: refused in round 2, when its kernel was capped at 128 registers and the second loop spilled; at today's 139 it
    // does not, and with the clamp form of the self-correction the clamp-free loop is worth +7 % at 1 048 576 frames:
    // 130.9 -> 140.0 M codewords/s, profiles/r03_kbench/kb4.txt, kb11_forms.txt)
    constexpr bool NOCAP_POSSIBLE = has_nocap_loop<CODE, T, IPT, LEAN>() && G == 1;
    auto cap_flag = [&]() LDPC_INLINE -> int & { return *reinterpret_cast<int *>(gbase + FLAG_OFF + 8); };

    // Byte offset, inside one block's M*SZ-byte LDS region, of the variable that check
    // i = S*NT + t of block B is wired to; `tb` is t*SZ.  Identity blocks rotate the whole
    // region, pi_k blocks move quarter j to quarter (theta_k + j) mod 4 and rotate inside it
    // (compact_parity_checks.rs:107-108).  Two VALU operations: add, and-or.
    constexpr int Q = M / 4, LQ = ilog2(Q);
    constexpr bool QUARTER_LITERAL = NT <= Q || JW >= 0; // the quarter of index S is a literal
    constexpr bool QUARTER_SCALAR = !QUARTER_LITERAL && Q >= 64;   // a wave never straddles quarters
    int rot_s[IPT][NB], base_s[IPT][NB];                 // wave-uniform (SGPR) constants, QUARTER_SCALAR only
    if constexpr (QUARTER_SCALAR) {
        const int jw = __builtin_amdgcn_readfirstlane(t >> LQ);
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                constexpr Block blk = P.blk[B];
                if constexpr (blk.kind == BLK_P) {
                    constexpr int K = blk.val;
                    const int j = S * (NT / Q) + jw;
                    const int phi = j == 0 ? phi_of(K, 0, M) : (j == 1 ? phi_of(K, 1, M) : (j == 2 ? phi_of(K, 2, M) : phi_of(K, 3, M)));
                    rot_s[S][B] = (phi + S * NT) * SZ;
                    constexpr int bias = INPLACE ? exch_slot(P, B) * BLK_BYTES : lds_bias(P, B, BLK_BYTES);
                    base_s[S][B] = (((theta_of(K) + j) & 3) << LQ) * SZ + bias;
                }
            });
        });
    }
    auto wire = [&](auto B_, auto S_, int tb) LDPC_INLINE -> int {
        constexpr int B = decltype(B_)::value, S = decltype(S_)::value;
        constexpr Block blk = P.blk[B];
        constexpr int bias = INPLACE ? exch_slot(P, B) * BLK_BYTES : lds_bias(P, B, BLK_BYTES);
        if constexpr (blk.kind == BLK_I) {
            return ((tb + (S * NT + blk.val) * SZ) & (M * SZ - 1)) | bias;
        } else if constexpr (QUARTER_LITERAL) {
            constexpr int j = JW >= 0 ? S * (NT / Q) + JW : (S * NT) / Q, K = blk.val;
            return ((tb + (phi_of(K, j, M) + S * NT) * SZ) & (Q * SZ - 1)) | ((((theta_of(K) + j) & 3) << LQ) * SZ + bias);
        } else if constexpr (QUARTER_SCALAR) {
            return ((tb + rot_s[S][B]) & (Q * SZ - 1)) | base_s[S][B];
        } else {
            return pi_dev<blk.val, M>(S * NT + tb / SZ, (S * NT + tb / SZ) >> LQ) * SZ + bias;
        }
    };

    // ---- state, all in registers -----------------------------------------------------------
    R u[IPT][NB];                 // check -> variable message per edge        (decoder.rs:375)
    R v[IPT][NB];                 // variable -> check message per edge        (decoder.rs:376)
    R va[IPT][NCOLS];             // marginals                                  (decoder.rs:377)
    R llr[IPT][NTX];              // channel LLRs, read from HBM once
    // f64 in-place mode: across iterations only the sign and the zero-ness of v matter (decoder.rs:422;
    // the magnitudes are consumed inside the check row), so v is two bit masks instead of 2 x 30 VGPRs
    constexpr bool VFLAGS = INPLACE && sizeof(R) == 8 && IPT * NB <= 64;
    using FW = std::conditional_t<(IPT * NB <= 32), unsigned, unsigned long long>;
    FW vneg = 0, vnz = 0;         // bit S*NB+B: v of that edge is negative / non-zero

    // Asynchronous global -> LDS copy of codeword `c`'s LLRs into the staging buffer.  Each wave
    // stages exactly the elements its own lanes will read back, so only the issuing wave's
    // vmcnt has to be waited for, never a barrier.  `stage` is a separate __shared__ object so
    // that the compiler does not order the decode loop's LDS traffic behind these copies.
    auto stage_issue = [&](uint32_t c) LDPC_INLINE {
        if constexpr (PF) {
            unsigned tu = (unsigned)t;
            asm volatile("" : "+v"(tu));
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                    constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
                    const T *src = (llrs + (size_t)c * N) + ((unsigned)(C * M + S * NT) + tu);   // scalar base + lane offset
                    char *dst = stage + (C * M + S * NT + (t & ~63)) * TSZ;           // wave-uniform
                    auto gsrc = (const __attribute__((address_space(1))) void *)src;
                    auto ldst = (__attribute__((address_space(3))) void *)dst;
                    // the size operand must be a literal
                    if constexpr (TSZ == 4) __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 0, 0);
                    else if constexpr (TSZ == 2) __builtin_amdgcn_global_load_lds(gsrc, ldst, 2, 0, 0);
                    else __builtin_amdgcn_global_load_lds(gsrc, ldst, 1, 0, 0);
                });
            });
        }
    };

    // Plain global loads of codeword `c`'s LLRs into registers.  Issued BEFORE the previous
    // codeword's hard-decision epilogue and this one's state initialisation, whose ~200
    // instructions cover the HBM latency; the registers are the ones `llr` vacates when a decode ends.
    T lraw[IPT][NTX];
    bool cap_wave = false;        // one-wave workgroups: the clamp vote of the current codeword (wave-uniform)
    // The register-lean kernels of the narrow LLR types keep their LLRs as PACKED raw values -- four i8 or two i16 per register,
    // three / five registers for TM5120's ten -- instead of re-reading them from L2 in every variable phase, as the lean f32
    // kernel must (it has no registers to spare): TM5120 i8 19.04 -> 20.64 (4 dB), 7.33 -> 7.84 M codewords/s (2 dB), no spills
    // (profiles/r03_kbench/kb3.txt).  Unpacking is a shift-and-sign-extend folded into the conversion to f32.
    constexpr bool PACKED_LLR = LEAN == 1 && sizeof(T) <= 2 && packed_llr_default<CODE, T>();
    constexpr int PER_REG = PACKED_LLR ? 4 / (int)sizeof(T) : 1, PK_BITS = 8 * (int)sizeof(T);
    unsigned llr_pk[IPT][(NTX + PER_REG - 1) / PER_REG];
    auto fetch_llrs = [&](uint32_t c) LDPC_INLINE {
        if constexpr (PACKED_LLR) {
            const uint32_t cc = c < batch ? c : batch - 1;
            using UT = std::conditional_t<sizeof(T) == 1, uint8_t, uint16_t>;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value;
                static_for<0, (NTX + PER_REG - 1) / PER_REG>([&](auto W_) LDPC_INLINE { llr_pk[S][decltype(W_)::value] = 0; });
                static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                    constexpr int C = decltype(C_)::value;
                    const unsigned b = (unsigned)(UT)(llrs + (size_t)cc * N)[(unsigned)(C * M + S * NT) + (unsigned)t];
                    llr_pk[S][C / PER_REG] |= b << (PK_BITS * (C % PER_REG));
                });
            });
        }
        if constexpr (LEAN) return;
        unsigned tu = (unsigned)t;
        asm volatile("" : "+v"(tu));
        const uint32_t cc = c < batch ? c : batch - 1;          // padding lanes of a last, partial group
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
                lraw[S][C] = (llrs + (size_t)cc * N)[(unsigned)(C * M + S * NT) + tu];
            });
        });
    };

    // Iteration 0 as a pass of its own (see check_phase): per kernel, peel_first_default()
    // (the NaN-blind first pass of TM1280 f32 is the one form of that kernel the peeled pass pays on: 75.4 -> 78.5 M codewords/s)
    constexpr bool PEEL_FIRST = (LDPC_PEEL_FIRST >= 0 ? LDPC_PEEL_FIRST != 0 : (peel_first_default<CODE, T, IPT>() || (NANPASS == 1 && CODE == TM1280))) && LEAN != 2;
    // float LLRs: canonicalise the finished marginals instead of the LLRs (Ops<float>::canon_late) -- the register-lean kernels,
    // which re-read their LLRs in every variable phase, and TM1280 f32
    constexpr bool NONAN = NANPASS == 1;         // first of two passes: NaN LLRs are only looked for (the codeword is marked), not handled
    constexpr bool REDO = NANPASS == 2;          // second pass: only the codewords the first one marked
    static_assert(NANPASS == 0 || (std::is_floating_point_v<T> && LEAN == 1 && G == 1 && !PF && GEO::WG > 64), "two-pass NaN handling: the register-lean float kernel, one multi-wave codeword per workgroup");
    constexpr bool LATE_CANON = std::is_floating_point_v<T> && (LEAN != 0 || (CODE == TM1280 && IPT == 1)) && !NONAN;
    // (Tried for the register-lean f32 kernel, which cannot afford even that -- 45 more spilled registers, TM5120 f32 17.1 ->
    // 14.4 M codewords/s: look for a NaN among a codeword's LLRs once, before the first pass, and run a second copy of the loop
    // that canonicalises only for codewords that have one.  The second copy alone costs more: 120 spilled registers, 13.3 M
    // codewords/s (LDPC_NANVOTE, kbench only; profiles/r03_kbench/kb16_nanvote.txt).  The vote survives as the MARK of the
    // two-pass form: the common copy is the whole first kernel, the canonicalising copy a second kernel.)
    constexpr bool NANVOTE = false;
    constexpr bool ZERO_FREE = PEEL_FIRST;
    auto begin_codeword = [&](bool staged) LDPC_INLINE {
        if constexpr (PF) {
            if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        int tb = t * SZ;
        unsigned tu = (unsigned)t;
        asm volatile("" : "+v"(tb), "+v"(tu));   // keep the address arithmetic inside the loop (see check_phase)
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            constexpr int S = decltype(S_)::value;
            const int i = S * NT + t;
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int B = decltype(B_)::value;
                u[S][B] = O::zero();                                               // decoder.rs:374
                v[S][B] = O::zero();
                vneg = 0; vnz = 0;
                constexpr int slot = exch_slot(P, B);
                if constexpr (slot >= 0 && !ZERO_FREE) {       // (a peeled first iteration never reads the slots)
                    constexpr int off = INPLACE ? 0 : lds_xu_off(P, slot, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                    lds_store(off + wire(B_, S_, tb), O::store(O::zero()));
                }
            });
            static_for<0, NCOLS>([&](auto C_) LDPC_INLINE { va[S][decltype(C_)::value] = O::zero(); });
            if constexpr (INPLACE) {
                // in-place mode keeps the exchanged columns' marginals only as sign words in LDS, written by the first
                // variable phase: a decode of ZERO iterations (marginals all zero, decoder.rs:374, :466-474) never runs one
                // and would hard-decide the previous codeword's words (found by the NaN goldens' max_iters = 0 leg, round 3)
                if (maxiters == 0)
                    static_for<0, NXC>([&](auto X_) LDPC_INLINE { *reinterpret_cast<int *>(gbase + hi_off(decltype(X_)::value) + i * 4) = 0; });
            }
            static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                constexpr int C = decltype(C_)::value;
                if (PF && staged) llr[S][C] = O::load(*reinterpret_cast<const T *>(stage + (C * M + i) * TSZ));
                else if constexpr (!LEAN) llr[S][C] = LATE_CANON ? O::keep_raw(lraw[S][C]) : O::load(lraw[S][C]);     // fetched by fetch_llrs()
                if constexpr (LLR_LDS) lds_store(LLR_OFF + (C * M + i) * SZ, O::store(llr[S][C]));      // (llr[][] dies after the range vote below)
            });
        });
        (void)tu;
        if (t < 2) flag_at(t) = 0;
        if constexpr (NOCAP_POSSIBLE) {
            bool big = false;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                    const R a = O::mag(llr[decltype(S_)::value][decltype(C_)::value]);
                    big |= !(a <= nocap_limit) || (a != 0.0f && a < 0x1p-20f);     // NaN counts as out of range
                });
            });
            if constexpr (GEO::WG == 64) cap_wave = __ballot(big) != 0;          // one wave = one codeword: no LDS word needed
            else if (__ballot(big) != 0 && (tid & 63) == 0) cap_flag() = 1;
        } else if constexpr (NANVOTE) {
            bool nn = false;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                    const T x = (llrs + (size_t)(live ? cw : 0) * N)[(unsigned)(decltype(C_)::value * M + decltype(S_)::value * NT) + (unsigned)t];
                    nn |= x != x;
                });
            });
            if (__ballot(nn) != 0 && (tid & 63) == 0) cap_flag() = 1;           // (the clamp-vote word serves as the NaN vote here)
        } else {
            if (t == 2) *reinterpret_cast<int *>(gbase + FLAG_OFF + 8) = 0;  // (the clamp-vote word, unused here)
        }
    };

    constexpr int CARRY_SET = LDPC_SELFCORR_CARRY >= 0 ? LDPC_SELFCORR_CARRY : selfcorr_carry_default<CODE>();
    constexpr bool CARRY = CARRY_SET == 2 || (CARRY_SET == 1 && !std::is_same_v<T, float>);
    // form of the self-correction (Ops::self_correct): the clamp forms where the values allow them -- integer messages
    // always, f32 only in the clamp-free copy of the loop (its codewords passed the range vote)
    constexpr int MED3 = FORM;
    constexpr int FORM_U = (MED3 != 0 && sizeof(T) <= 2) ? MED3 : (CARRY ? 1 : 0);
    constexpr int FORM_B = MED3 != 0 ? MED3 : (CARRY ? 1 : 0);
    auto edge_update = [&](auto S_, auto B_, R x, R uu, auto BND_) LDPC_INLINE {
        constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
        constexpr bool BND = decltype(BND_)::value != 0 && G == 1;
        const R nv = O::sub_nv(x, uu);                                                 // :421
        // keep nv if its sign equals the old v's or the old v is zero, else zero it (:422-425)
        const R nw = BND ? O::template self_correct_b<true, FORM_B>(nv, v[S][B]) : O::template self_correct<FORM_U>(nv, v[S][B]);
        v[S][B] = nw;
    };
    // the part of the check update that needs no exchanged data: the LOCAL edges
    // The first LOCAL_IN_VAR of them (index-major rank) are done at the END of the variable phase -- LDS-bound,
    // the VALU idles there -- the rest at the start of the check phase, where they cover the latency of the
    // marginal reads (TM8192 pair kernel: +5 %; per-code values measured with tools/kbench.hip).
    constexpr int LOCAL_IN_VAR = LDPC_LOCAL_IN_VAR >= 0 ? LDPC_LOCAL_IN_VAR : local_in_var_default<CODE, T, IPT, LEAN>();
    auto local_edges = [&](auto EARLY_, auto BND_) LDPC_INLINE {
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                if constexpr (exch_slot(P, B) < 0 && (local_edge_rank(P, S, B) < LOCAL_IN_VAR) == (decltype(EARLY_)::value != 0))
                    edge_update(S_, B_, va[S][P.blk[B].col], u[S][B], BND_);
            });
        });
    };
    auto check_local = [&](auto BND_) LDPC_INLINE { local_edges(IC<0>{}, BND_); };

    // bit pattern (sign in bit 31) of the marginal of variable (S, C) of this thread
    auto marginal_bits = [&](auto S_, auto C_) LDPC_INLINE -> int {
        constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
        constexpr int cs = col_slot(P, C);
        if constexpr (INPLACE && cs >= 0) return *reinterpret_cast<const int *>(gbase + hi_off(cs) + (S * NT + t) * 4);
        else return O::bits(va[S][C]);
    };

    // One iteration of message passing for this thread's indices: the two phases below.
    // FIRST_: iteration 0 of a codeword, peeled by kernels with PEEL_FIRST -- every u is zero, so the marginals are the
    // LLRs and nothing is read from LDS
    // CANON_ (NANVOTE kernels): this codeword has a NaN LLR -- canonicalise the finished marginals; 0 = the common copy
    auto variable_phase = [&](auto FIRST_, auto CANON_) LDPC_INLINE {
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        constexpr bool CANON = LATE_CANON && (!NANVOTE || decltype(CANON_)::value != 0);
        // marginals (decoder.rs:382-383, :408)
        int tv = t;
        if constexpr (INPLACE) asm volatile("" : "+v"(tv));     // keep the (large-offset) LDS addresses out of loop-carried VGPRs
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            constexpr int S = decltype(S_)::value;
            const int i = S * NT + tv;
            if constexpr (S == 0) LDPC_SETPRIO(LDPC_PRIO_VAR);
            else if constexpr (S == IPT / 2) LDPC_SETPRIO(0);
            static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
                constexpr int C = decltype(C_)::value;
                R acc = O::zero();
                if constexpr (C < NTX) {
                    if constexpr (PACKED_LLR) acc = O::load((T)(llr_pk[S][C / PER_REG] >> (PK_BITS * (C % PER_REG))));
                    else if constexpr (LEAN) {
                        const T x = (llrs + (size_t)(live ? cw : 0) * N)[(unsigned)(C * M) + (unsigned)i];
                        // (first of two NaN passes: the raw value here and `+ 0.0` on the finished marginal, as canon_late does it --
                        // with the `+ 0.0` up here the lean TM5120 kernel spills 53 registers instead of 8)
                        acc = (CANON || NONAN) ? O::keep_raw(x) : (NANVOTE ? O::load_nonan(x) : O::load(x));
                        
                    }
                    else if constexpr (LLR_LDS) acc = O::from_lds(lds_load(LLR_OFF + (C * M + i) * SZ));
                    else acc = llr[S][C];
                }
                if constexpr (INPLACE) {
                    R ue[NB];                                   // the u of this column's edges, in list order
                    static_for<0, NB>([&](auto B_) LDPC_INLINE {
                        constexpr int B = decltype(B_)::value;
                        if constexpr (P.blk[B].col == C) {
                            constexpr int slot = exch_slot(P, B);
                            if constexpr (slot >= 0) ue[B] = O::from_lds(lds_load(slot * BLK_BYTES + i * SZ));
                            else ue[B] = u[S][B];
                            acc = O::add(acc, ue[B]);                                  // :408
                        }
                    });
                    if constexpr (CANON && C < NTX) acc = O::canon_late(acc);        // (see Ops<float>::canon_late)
                    else if constexpr (NONAN && C < NTX) acc = O::load_nonan(acc);
                    if constexpr (col_slot(P, C) < 0) va[S][C] = acc;               // exchanged columns: only the sign word is kept (hi array)
                    static_for<0, NB>([&](auto B_) LDPC_INLINE {                       // nv = va - u (:421), in place
                        constexpr int B = decltype(B_)::value;
                        if constexpr (P.blk[B].col == C) {
                            constexpr int slot = exch_slot(P, B);
                            const R nv = O::sub(acc, ue[B]);
                            if constexpr (slot >= 0) lds_store(slot * BLK_BYTES + i * SZ, O::store(nv));
                            else u[S][B] = nv;
                        }
                    });
                    constexpr int cs = col_slot(P, C);
                    if constexpr (cs >= 0) *reinterpret_cast<int *>(gbase + hi_off(cs) + i * 4) = O::bits(acc);
                } else {
                if constexpr (!FIRST)
                static_for<0, NB>([&](auto B_) LDPC_INLINE {
                    constexpr int B = decltype(B_)::value;
                    if constexpr (P.blk[B].col == C) {
                        constexpr int slot = exch_slot(P, B);
                        if constexpr (slot >= 0) {
                            constexpr int off = lds_xu_off(P, slot, BLK_BYTES);
                            acc = O::add(acc, O::from_lds(lds_load(off + i * SZ)));
                        }
                        else acc = O::add(acc, u[S][B]);
                    }
                });
                if constexpr (CANON && C < NTX) acc = O::canon_late(acc);          // (see Ops<float>::canon_late)
                else if constexpr (NONAN && C < NTX) acc = O::load_nonan(acc);
                va[S][C] = acc;
                constexpr int cs = col_slot(P, C);
                if constexpr (cs >= 0) {
                    constexpr int off = lds_xva_off(P, cs, BLK_BYTES);
                    lds_store(off + i * SZ, O::store(acc));
                }
                }
            });
        });
        if constexpr (LEAN == 0 && !FIRST) local_edges(IC<1>{}, IC<0>{});      // (kernels that move local edges here have no bounded mode; a peeled first pass sets every v in its check phase)
    };

    // Codewords that live inside ONE wave (the TC codes: 16 / 32 / 64 threads each) get their verdict without a
    // barrier: the parity of every owned check (decoder.rs:445-447) is evaluated as soon as the exchanged marginals
    // have arrived, one ballot tells every lane whether all checks of its codeword hold (decoder.rs:453), and if
    // they do the rest of the check phase -- the updates, minima and messages of an iteration whose result nobody
    // reads -- is skipped.  A success that takes k iterations then costs k + 0.35 passes instead of k + 1: +10-20 % at
    // 5 dB (2 iterations), +1-12 % at 3 dB (min_waves_per_simd() has the table); config 2 (TC512 f32 at 2 dB, 15
    // iterations, 29 % failures) 109.5 -> 106.9.  Multi-wave codewords need one more barrier per iteration for this:
    // WG_VERDICT below.
    constexpr bool WAVE_VERDICT = LDPC_WAVE_VERDICT && GEO::WG == 64 && LEAN == 0;
    // The same for codewords of several waves needs a third barrier per iteration (parity -> flag -> barrier -> read).
    // Where several workgroups share a CU the barrier hides behind the others and the skipped half pass is a net gain:
    // TM1536 f32 59.0 -> 62.5 (3 dB; 31.6 -> 31.9 at 2 dB where most frames fail), i8 55.5 -> 57.6; TM1280 f32 67.8 ->
    // 70.3 (its i8 / i16 kernels spill 23 registers with it: 65.2 -> 58.7; they run the lean kernel now); i32 +5-7 %, f64
    // +1-5 % on both codes.  Not where a workgroup fills a CU or the
    // iteration count is high: TM2048 -1 % (2.5 dB) / -5 % (config 3), TM6144 -2.4 %; the lean TM5120 kernel would
    // have to read its 39 marginals twice.
    constexpr int WG_VERDICT_SET = LDPC_WG_VERDICT >= 0 ? LDPC_WG_VERDICT
                                 : (CODE == TM1536 || (CODE == TM1280 && !(sizeof(T) <= 2))) ? 1 : 0;
    constexpr bool WG_VERDICT = WG_VERDICT_SET != 0 && !WAVE_VERDICT && G == 1 && IPT == 1 && LEAN == 0;
    // The register-lean check phase cannot hold its marginals between a parity pass and the updates (that is what makes it
    // lean), but it can split the other way: pass A = requests, edge updates, parities and sign words of ALL rows; vote
    // through the flag and a third barrier; pass B = exclusive minima, next u and their stores, skipped on success (it reads
    // only v, which is in registers).  LDPC_LEAN_VERDICT: measured in round 3 (see lean_verdict_default()).
    constexpr bool LEAN_VERDICT = (LDPC_LEAN_VERDICT >= 0 ? LDPC_LEAN_VERDICT != 0 : lean_verdict_default<CODE, T>()) && LEAN == 1 && G == 1;
    constexpr bool IN_PHASE_VERDICT = WAVE_VERDICT || WG_VERDICT || LEAN_VERDICT;
    // Iteration 0 peeled (PEEL_FIRST; at high SNR a decode is two or three passes, and the first one is cheaper than
    // the rest): u = 0 and v = 0 make every new v the marginal itself (decoder.rs:421-425
    // with u == 0 and v == 0: x - 0, kept), so the pass needs no LDS reads in its variable phase and no
    // subtract / test / select per edge in its check phase, and the exchange slots need no zeroing.
    auto check_phase = [&](uint32_t it, auto CAP_, auto FIRST_) LDPC_INLINE -> bool {
        constexpr bool CAP = decltype(CAP_)::value != 0;
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        // decoder.rs:414-450, and :391-405 of the NEXT iteration
        int par_any = 0;          // bit 31 set if any owned check has odd parity
        // LDS addresses of the exchanged edges are two VALU ops each from `tb`; making `tb`
        // opaque per iteration stops the compiler from hoisting all of them into VGPRs that
        // then stay live across the whole loop (which costs more in spills than it saves).
        int tb = t * SZ;
        asm volatile("" : "+v"(tb));
        LDPC_SETPRIO(3);
        // Order of work, chosen so that LDS latency is covered by arithmetic: (1) request the
        // marginals of all exchanged edges, (2) update the LOCAL edges (their marginals are in
        // registers), (3) update the exchanged edges as their data arrives, (4) per check:
        // exclusive minima, signs, next u, and the LDS stores of the exchanged u.
        R xs[IPT][NB];
        int ad[IPT][NB];
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                constexpr int slot = exch_slot(P, B);
                if constexpr (slot >= 0) {
                    constexpr int cs = col_slot(P, P.blk[B].col);
                    constexpr int off = lds_xva_off(P, cs, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                    ad[S][B] = wire(B_, S_, tb);
                    xs[S][B] = O::from_lds(lds_load(off + ad[S][B]));
                }
            });
        });
        __builtin_amdgcn_sched_barrier(0);    // keep the requests ahead of the local-edge work
        if constexpr (WAVE_VERDICT) {     // (evaluating it after the local edges instead, which would cover the LDS latency, measures the same)
            int pe = 0;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value;
                static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                    constexpr int Rw = decltype(R_)::value;
                    constexpr int D = row_degree(P, Rw);
                    int xw[D];
                    static_for<0, D>([&](auto J_) LDPC_INLINE {
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        if constexpr (exch_slot(P, B) >= 0) xw[J] = O::bits(xs[S][B]);     // :445-447
                        else xw[J] = O::bits(va[S][P.blk[B].col]);
                    });
                    pe |= xor_reduce<D>(xw);
                });
            });
            const unsigned long long odd = __ballot(pe < 0);                            // lanes with an unsatisfied check
            constexpr unsigned long long GROUP = NT >= 64 ? ~0ull : ((1ull << (NT & 63)) - 1ull);
            if (((odd >> (grp * NT)) & GROUP) == 0) return true;                        // :453
        }
        if constexpr (WG_VERDICT) {
            int pe = 0;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value;
                static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                    constexpr int Rw = decltype(R_)::value;
                    constexpr int D = row_degree(P, Rw);
                    int xw[D];
                    static_for<0, D>([&](auto J_) LDPC_INLINE {
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        if constexpr (exch_slot(P, B) >= 0) xw[J] = O::bits(xs[S][B]);
                        else xw[J] = O::bits(va[S][P.blk[B].col]);
                    });
                    pe |= xor_reduce<D>(xw);
                });
            });
            if (__ballot(pe < 0) != 0 && (tid & 63) == 0) flag_at(it) = 1;
            LDPC_SYNC();
            if (flag_at(it) == 0) return true;
        }
        // bounded mode = the clamp-free copy of the loop: its codewords passed the LLR range vote
        constexpr int BND = (!CAP && NOCAP_POSSIBLE && LOCAL_IN_VAR == 0) ? 1 : 0;
        if constexpr (FIRST) {
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {                              // (2) + (3) with u == 0, v == 0
                static_for<0, NB>([&](auto B_) LDPC_INLINE {
                    constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                    if constexpr (exch_slot(P, B) >= 0) v[S][B] = xs[S][B];
                    else v[S][B] = va[S][P.blk[B].col];
                });
            });
        } else {
        check_local(IC<BND>{});                                                        // (2)
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {                                  // (3)
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                if constexpr (exch_slot(P, B) >= 0) edge_update(S_, B_, xs[S][B], u[S][B], IC<BND>{});
            });
        });
        }
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {                                  // (4)
            constexpr int S = decltype(S_)::value;
            static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                constexpr int Rw = decltype(R_)::value;
                constexpr int D = row_degree(P, Rw);
                {   // priority of this (index, row) step: LDPC_PRIO_ROWS lists it for the LAST 6 steps of the phase
                    constexpr int prio_rows[6] = LDPC_PRIO_ROWS;
                    constexpr int step = S * NROWS + Rw, nsteps = IPT * NROWS;
                    constexpr int k = step - (nsteps - 6);
                    constexpr int now = k >= 0 ? prio_rows[k] : 3, before = (step > 0 && k >= 1) ? prio_rows[k - 1] : 3;
                    if constexpr (now != before) LDPC_SETPRIO(now);
                }
                R a[D], e[D];
                int sr[D], xw[D];
                static_for<0, D>([&](auto J_) LDPC_INLINE {
                    constexpr int J = decltype(J_)::value;
                    constexpr int B = row_block(P, Rw, J);
                    a[J] = v[S][B];                                                    // magnitude taken in exclusive_min
                    sr[J] = O::sign_word(v[S][B]);                                     // sign word, :439-441
                    if constexpr (exch_slot(P, B) >= 0) xw[J] = O::bits(xs[S][B]);     // :445-447
                    else xw[J] = O::bits(va[S][P.blk[B].col]);
                });
                const int sgn = xor_reduce<D>(sr), par = (WAVE_VERDICT || WG_VERDICT) ? 0 : xor_reduce<D>(xw);
                exclusive_min<O, D, true, CAP>(a, e);                                       // :391-395, :430-435
                static_for<0, D>([&](auto J_) LDPC_INLINE {
                    constexpr int J = decltype(J_)::value;
                    constexpr int B = row_block(P, Rw, J);
                    const R un = O::apply_sign(e[J], sgn, sr[J]);                      // :398-405
                    u[S][B] = un;
                    constexpr int slot = exch_slot(P, B);
                    if constexpr (slot >= 0) {
                        constexpr int off = lds_xu_off(P, slot, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                        lds_store(off + ad[S][B], O::store(un));
                    }
                });
                par_any |= par;
            });
        });
        if constexpr (WAVE_VERDICT || WG_VERDICT) return false;       // (some check of this codeword is unsatisfied: next iteration)
        if (par_any < 0) flag_at(it) = 1;
        return false;
    };

    // Register-lean check phase (LEAN): one check row at a time, its edges in chunks of six; the u
    // of an exchanged edge is read back from the LDS slot it was stored to, nothing per edge but v
    // (and the u of local edges) stays live between chunks; addresses are recomputed for the store.
    auto check_phase_lean = [&](uint32_t it, auto FIRST_) LDPC_INLINE -> bool {
        constexpr bool FIRST = decltype(FIRST_)::value != 0;      // iteration 0 peeled: u == 0, v == 0 (see check_phase)
        int par_any = 0;
        int tb = t * SZ;
        asm volatile("" : "+v"(tb));
        int sgn_row[IPT][NROWS];                                  // (LEAN_VERDICT: the rows' sign words, carried from pass A to pass B)
        // pass B of one row: exclusive minima, next u, stores (decoder.rs:391-405 of the next iteration)
        auto finish_row = [&](auto S_, auto R_, int sgn) LDPC_INLINE {
            constexpr int S = decltype(S_)::value, Rw = decltype(R_)::value;
            constexpr int D = row_degree(P, Rw);
            R a[D], e[D];
            static_for<0, D>([&](auto J_) LDPC_INLINE {
                constexpr int J = decltype(J_)::value, B = row_block(P, Rw, J);
                a[J] = v[S][B];
            });
            exclusive_min<O, D, true>(a, e);                                           // :391-395, :430-435
            static_for<0, D>([&](auto J_) LDPC_INLINE {
                constexpr int J = decltype(J_)::value;
                constexpr int B = row_block(P, Rw, J);
                const R un = O::apply_sign(e[J], sgn, O::sign_word(v[S][B]));               // :398-405
                constexpr int slot = exch_slot(P, B);
                if constexpr (slot >= 0) {
                    constexpr int off = lds_xu_off(P, slot, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                    lds_store(off + wire(IC<B>{}, S_, tb), O::store(un));
                } else {
                    u[S][B] = un;
                }
            });
        };
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            constexpr int S = decltype(S_)::value;
            static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                constexpr int Rw = decltype(R_)::value;
                constexpr int D = row_degree(P, Rw);
                constexpr int CH = LDPC_LEAN_CH, NCH = (D + CH - 1) / CH;
                {
                    constexpr int prio_rows[6] = LDPC_PRIO_ROWS_LEAN;
                    constexpr int step = S * NROWS + Rw, nsteps = IPT * NROWS;
                    constexpr int k = step - (nsteps - 6);
                    constexpr int now = k >= 0 ? prio_rows[k] : 3, before = (step > 0 && k >= 1) ? prio_rows[k - 1] : -1;
                    if constexpr (now != before) LDPC_SETPRIO(now);
                }
                // par: XOR of the marginals' words (bit 31: the check's parity, :445-447); sgnw: XOR of the new v's WHOLE words --
                // bit 31 is the row's sign product (:439-441), masked once per row instead of once per edge.  Both take two
                // words per v_bitop3_b32: 4.4 -> 3.5 sign / parity instructions per edge; TM5120 i8 21.12 -> 21.90 M codewords/s at
                // 4 dB, 7.97 -> 8.29 at 2 dB (profiles/r03_kbench/kb18_form6.txt).
                int par = 0, sgnw = 0;
                static_for<0, NCH>([&](auto K_) LDPC_INLINE {
                    constexpr int J0 = decltype(K_)::value * CH, J1 = J0 + CH < D ? J0 + CH : D;
                    R xr[CH], ur[CH];
                    int xw[CH], vw[CH];
                    static_for<J0, J1>([&](auto J_) LDPC_INLINE {                      // requests
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        constexpr int slot = exch_slot(P, B);
                        if constexpr (slot >= 0) {
                            constexpr int cs = col_slot(P, P.blk[B].col);
                            constexpr int offx = lds_xva_off(P, cs, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                            constexpr int offu = lds_xu_off(P, slot, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                            const int adr = wire(IC<B>{}, S_, tb);
                            xr[J - J0] = O::from_lds(lds_load(offx + adr));
                            if constexpr (!FIRST) ur[J - J0] = O::from_lds(lds_load(offu + adr));
                        } else {
                            xr[J - J0] = va[S][P.blk[B].col];
                            if constexpr (!FIRST) ur[J - J0] = u[S][B];
                        }
                    });
                    static_for<J0, J1>([&](auto J_) LDPC_INLINE {
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        if constexpr (FIRST) v[S][B] = xr[J - J0];
                        else edge_update(S_, IC<B>{}, xr[J - J0], ur[J - J0], IC<0>{});     // :421-425
                        xw[J - J0] = O::bits(xr[J - J0]);
                        vw[J - J0] = O::bits(v[S][B]);
                    });
                    par = xor_into<J1 - J0>(par, xw);
                    sgnw = xor_into<J1 - J0>(sgnw, vw);
                });
                static_assert(O::SIGN_WORD_IS_BIT31_ONLY, "the register-lean check phase forms sign words from raw bits");
                const int sgn = sgnw & (int)0x80000000;
                if constexpr (LEAN_VERDICT) sgn_row[S][Rw] = sgn;
                else finish_row(S_, R_, sgn);
                par_any |= par;
            });
        });
        if constexpr (LEAN_VERDICT) {
            if (__ballot(par_any < 0) != 0 && (tid & 63) == 0) flag_at(it) = 1;
            LDPC_SYNC();
            if (flag_at(it) == 0) return true;                                         // :453: every check of the codeword holds
            tb = t * SZ;
            asm volatile("" : "+v"(tb));          // pass B recomputes its addresses: none may stay live across the vote
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NROWS>([&](auto R_) LDPC_INLINE { finish_row(S_, R_, sgn_row[decltype(S_)::value][decltype(R_)::value]); });
            });
            return false;
        } else {
            if (par_any < 0) flag_at(it) = 1;
            return false;
        }
    };

    // In-place check phase (LEAN == 2): as the lean one, but an exchanged edge's slot already holds
    // nv = va - u (written by the variable thread), its marginal's sign word comes from the hi array,
    // and a local edge's u register holds its nv.
    auto check_phase_inplace = [&](uint32_t it) LDPC_INLINE {
        int par_any = 0;
        int tb = t * SZ;
        asm volatile("" : "+v"(tb));
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            constexpr int S = decltype(S_)::value;
            static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                constexpr int Rw = decltype(R_)::value;
                constexpr int D = row_degree(P, Rw);
                constexpr int CH = LDPC_LEAN_CH, NCH = (D + CH - 1) / CH;
                int par = 0, sgn = 0;
                R a[D], e[D];                                   // this row's new v, its exclusive minima
                static_for<0, NCH>([&](auto K_) LDPC_INLINE {
                    constexpr int J0 = decltype(K_)::value * CH, J1 = J0 + CH < D ? J0 + CH : D;
                    R nr[CH];
                    int xw[CH];
                    static_for<J0, J1>([&](auto J_) LDPC_INLINE {
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        constexpr int slot = exch_slot(P, B);
                        if constexpr (slot >= 0) {
                            constexpr int cs = col_slot(P, P.blk[B].col);
                            const int adr = wire(IC<B>{}, S_, tb);                     // slot base + 8 * position
                            nr[J - J0] = O::from_lds(lds_load(adr));
                            xw[J - J0] = *reinterpret_cast<const int *>(gbase + hi_off(cs) + (adr - slot * BLK_BYTES) / (SZ / 4));
                        } else {
                            nr[J - J0] = u[S][B];
                            xw[J - J0] = marginal_bits(S_, IC<P.blk[B].col>{});
                        }
                    });
                    static_for<J0, J1>([&](auto J_) LDPC_INLINE {
                        constexpr int J = decltype(J_)::value;
                        constexpr int B = row_block(P, Rw, J);
                        if constexpr (VFLAGS) {
                            constexpr FW bit = (FW)1 << (S * NB + B);
                            const R nv = nr[J - J0];
                            const bool drop = (vnz & bit) && ((nv < O::zero()) != ((vneg & bit) != 0));   // :422
                            const R nw = drop ? O::zero() : nv;
                            vneg = nw < O::zero() ? (vneg | bit) : (vneg & ~bit);
                            vnz = nw != O::zero() ? (vnz | bit) : (vnz & ~bit);
                            a[J] = nw;
                        } else {
                            v[S][B] = O::template self_correct<false>(nr[J - J0], v[S][B]);   // :422-425
                            a[J] = v[S][B];
                        }
                        par ^= xw[J - J0];                                             // :445-447
                        static_assert(O::SIGN_WORD_IS_BIT31_ONLY, "the in-place check phase forms sign words from raw bits");
                        sgn ^= O::bits(a[J]) & (int)0x80000000;                        // :439-441
                    });
                });
                exclusive_min<O, D, true>(a, e);                                       // :391-395, :430-435
                static_for<0, D>([&](auto J_) LDPC_INLINE {
                    constexpr int J = decltype(J_)::value;
                    constexpr int B = row_block(P, Rw, J);
                    const R un = O::apply_sign(e[J], sgn, O::bits(a[J]) & (int)0x80000000);      // :398-405
                    if constexpr (exch_slot(P, B) >= 0) lds_store(wire(IC<B>{}, S_, tb), O::store(un));
                    else u[S][B] = un;
                });
                par_any |= par;
            });
        });
        if (par_any < 0) flag_at(it) = 1;
    };

    // DYNAMIC DISTRIBUTION (claim != nullptr).  Iteration counts are data dependent (3 ... max_iters), so equal shares
    // of the batch are unequal shares of the work: the reference's harness lets every worker pull the next trial until
    // the job is done (perftest/src/main.rs:39-45), and so do the workgroups here.  `*claim` is the launch's queue head,
    // zero between launches.  At the START of a chunk's first decode one lane draws a ticket (one atomic per chunk); the
    // chunk it names, gridDim.x + ticket, is the workgroup's NEXT one, so the atomic's round trip is hidden behind the
    // first pass of the decode and the value is collected from the returning register afterwards (collect_claim: into an
    // SGPR for one-wave workgroups, through one LDS word otherwise -- a VGPR held across the whole decode would cost the
    // large kernels their occupancy).  Every decode of the launch draws at most once and the draws of a launch number
    // exactly n_chunks, so the holder of ticket n_chunks - 1 knows it drew last and puts the head back to zero: no
    // memset between launches.  The launcher hands out one queue head per stream (launches of a stream run in order).
    const bool dyn = claim != nullptr && maxiters != 0;       // (a decode of zero iterations has no first pass to hide behind)
    const uint32_t CLAIM_K = dyn && claim_k != 0 ? claim_k : 1u;          // groups per draw, chosen by the launcher (a fixed stride goes group by group)
    const uint32_t n_chunks = (n_groups + CLAIM_K - 1) / CLAIM_K;
    uint32_t ticket = 0;                                      // lane 0 of wave 0: the returning atomic
    uint32_t next_chunk = 0;                                  // one-wave workgroups: the collected claim (wave-uniform)
    int *const next_word = reinterpret_cast<int *>(lds + FLAG_OFF + 12);
    bool fresh = true;                                        // first decode of a chunk (wave-uniform)
    // (kernels without a peeled first pass would have to carry the ticket into their iteration loop -- TM1280 f32 then
    // spills inside it -- so they draw at the END of a chunk's first decode instead and wait for the answer there)
    constexpr bool CLAIM_AHEAD = PEEL_FIRST;
    auto collect_claim = [&]() LDPC_INLINE {
        if (dyn && fresh) {
            if constexpr (GEO::WG == 64) {
                next_chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gridDim.x + ticket));
                if (tid == 0 && ticket == n_chunks - 1) __hip_atomic_store(claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (tid == 0) {
                *next_word = (int)(gridDim.x + ticket);
                if (ticket == n_chunks - 1) __hip_atomic_store(claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    if constexpr ((NOCAP_POSSIBLE && GEO::WG != 64) || NANVOTE || NONAN) { if (t == 0) cap_flag() = 0; LDPC_SYNC(); }
    // Second NaN pass: workgroup b looks through the groups b, b + gridDim.x, ... for the ones the first pass marked, 64 at a
    // time (one load per lane, one ballot; every wave of the workgroup computes the same wave-uniform answer).  A launch
    // without a NaN is a few such loads per workgroup.
    auto next_marked = [&](uint32_t g0) LDPC_INLINE -> uint32_t {
        const uint32_t stride = gridDim.x;
        while (g0 < n_groups) {
            const uint32_t cand = g0 + (uint32_t)(tid & 63) * stride;           // (n_groups < 2^31 and 64 * stride < 2^20: no wrap)
            const bool hit = cand < n_groups && iters_out[cand] == NAN_MARK;
            const unsigned long long m = __ballot(hit);
            if (m != 0) return g0 + (uint32_t)__builtin_ctzll(m) * stride;
            g0 += 64u * stride;
        }
        return n_groups;
    };
    (void)next_marked;
    if constexpr (!REDO) { if (blockIdx.x * CLAIM_K < n_groups) fetch_llrs(G == 1 ? blockIdx.x * CLAIM_K : blockIdx.x * CLAIM_K * G + grp); }
    uint32_t chunk = blockIdx.x, g = chunk * CLAIM_K, g_end = g + CLAIM_K;       // [g, g_end): the rest of the current chunk
    if constexpr (REDO) {                        // (the launcher gives the second pass no queue: CLAIM_K = 1)
        chunk = next_marked(blockIdx.x); g = chunk; g_end = g + 1;
        if (g < n_groups) fetch_llrs(g);
    }
    for (uint32_t first = 1; g < n_groups; first = 0) {
    cw = G == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)g) : g * G + grp;
    live = cw < batch;
    if constexpr (CLAIM_AHEAD) {
        if (dyn && fresh && tid == 0) ticket = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!first) fetch_llrs(cw);    // (issuing them before the previous epilogue measured 1.5 % slower here: register allocation)
    begin_codeword(!first);
    if constexpr (PF) {
        // the staged LLRs are in registers (the loads above were waited for by their use in
        // O::load's consumers only after lgkmcnt; make that explicit), now refill the stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!dyn && CLAIM_K == 1 && g + gridDim.x < n_groups) stage_issue((uint32_t)__builtin_amdgcn_readfirstlane((int)(g + gridDim.x)));
    }

    // Codewords that share a wave (G > 1) finish at different iterations: a finished one
    // simply stops updating (its lanes are masked off) until the whole wave is done.
    bool done = false, ok = false;
    uint32_t iters = maxiters;
    // the iterations; one copy of the loop per clamp mode (two check phases inside one loop spill)
    auto iterate = [&](auto CAP_) LDPC_INLINE {
    uint32_t it0 = 0;
    if constexpr (PEEL_FIRST) {
        if (maxiters == 0) done = true;
        if constexpr (G == 1) { if (done) return; }
        else { if (__all(done)) return; }
        if (G == 1 || !done) variable_phase(IC<1>{}, CAP_);
        LDPC_SYNC();
        if (G == 1 || !done) {
            if constexpr (LEAN == 1) { if (check_phase_lean(0u, IC<1>{})) { done = true; ok = true; iters = 0; } }
            else if (check_phase(0u, CAP_, IC<1>{})) { done = true; ok = true; iters = 0; }
        }
        collect_claim();
        if constexpr (G == 1) { if (done) return; }
        it0 = 1;
    }
    for (uint32_t it = it0;; ++it) {
        if (it > 0) LDPC_SYNC();  // u of the exchanged blocks and the parity vote are visible (iteration 0: barrier below)
        // verdict on the previous iteration (decoder.rs:453-463, :466-474)
        if (!done) {
            if constexpr (!IN_PHASE_VERDICT) {
                if (it > 0 && flag_at(it - 1) == 0) { done = true; ok = true; iters = it - 1; }
            }
            if (!done && it == maxiters) { done = true; }
        }
        if constexpr (G == 1) { if (done) break; }
        else { if (__all(done)) break; }

        if (G == 1 || !done) variable_phase(IC<0>{}, CAP_);
        LDPC_SYNC();
        if constexpr (!WAVE_VERDICT) { if (it > 0 && t == 0) flag_at(it - 1) = 0; }
        if (G == 1 || !done) {
            if constexpr (LEAN == 2) check_phase_inplace(it);
            else if constexpr (LEAN == 1) { if (check_phase_lean(it, IC<0>{})) { done = true; ok = true; iters = it; } }
            else if (check_phase(it, CAP_, IC<0>{})) { done = true; ok = true; iters = it; }      // (wave verdict, decoder.rs:453-463)
        }
        if constexpr (IN_PHASE_VERDICT && G == 1) { if (done) break; }
    }
    };
    LDPC_SYNC();                  // the zeroed exchange slots, the flags and the clamp vote are visible
    if constexpr (NOCAP_POSSIBLE || NANVOTE) {
        if (GEO::WG == 64 ? cap_wave : __builtin_amdgcn_readfirstlane(cap_flag()) != 0) iterate(IC<1>{});
        else iterate(IC<0>{});
    } else {
        iterate(IC<1>{});
    }

    // ---- hard decision of the marginals, MSB first (decoder.rs:455-461 / :467-473) ------------
    // First of two NaN passes: where is a NaN LLR noticed?  At the END, in the marginals: without the NaN mapping a variable's
    // marginal is NaN after every variable phase exactly if its LLR is (NaN + u stays NaN; nothing else makes one: every u is a
    // finite +-min(|v|..., maxval) -- v_min ignores NaN operands -- and a finite or infinite LLR plus finite numbers is no NaN),
    // and the marginals are live here anyway.  (Testing the LLRs where they are loaded -- at the start of the codeword, or in the
    // lean kernel's first variable phase -- keeps them alive side by side and costs that kernel 45-70 spilled registers.)
    bool nan_seen = false;
    (void)nan_seen;
    if constexpr (NT >= 64) {
        unsigned long long w[IPT][NCOLS];
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
                const unsigned long long bits = __ballot(marginal_bits(S_, C_) < 0);   // bit l = lane l
                if constexpr (NONAN && C < NTX) nan_seen |= va[S][C] != va[S][C];
                const unsigned lo = __builtin_bswap32(__builtin_bitreverse32((unsigned)bits));
                const unsigned hi = __builtin_bswap32(__builtin_bitreverse32((unsigned)(bits >> 32)));
                w[S][C] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            });
        });
        if ((tid & 63) == 0) {
            uint8_t *dst = (output + (size_t)cw * GEO::OUT_LEN) + t / 8;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
                    constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
                    *reinterpret_cast<unsigned long long *>(dst + (C * M + S * NT) / 8) = w[S][C];
                });
            });
        }
    } else {
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, C = decltype(C_)::value;
                const unsigned long long bits = __ballot(marginal_bits(S_, C_) < 0);
                const unsigned b8 = (unsigned)(bits >> (tid & 56)) & 0xFFu;
                if ((tid & 7) == 0 && live)
                    output[(size_t)cw * GEO::OUT_LEN + (C * M + S * NT + t) / 8] = (uint8_t)(__builtin_bitreverse32(b8) >> 24);
            });
        });
    }
    if constexpr (NONAN) {
        static_assert(!NOCAP_POSSIBLE && !NANVOTE && NT >= 64, "first NaN pass: the clamp-vote word is the mark");
        if (maxiters != 0 && __ballot(nan_seen) != 0 && (tid & 63) == 0) cap_flag() = 1;       // (zero iterations: no marginal was ever computed, and none depends on an LLR)
    } else {
        if (t == 0 && live) { iters_out[cw] = iters; success_out[cw] = ok ? 1 : 0; }
    }
    if constexpr ((NOCAP_POSSIBLE && GEO::WG != 64) || NANVOTE) { if (t == 0) cap_flag() = 0; }
    if constexpr (!CLAIM_AHEAD) {
        if (dyn && fresh && tid == 0) ticket = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        collect_claim();
    }
    LDPC_SYNC();                  // every wave is done with the flags before the next codeword resets them
    if constexpr (NONAN) {        // the waves' marks are in: results or mark, and the word is clear again long before the next epilogue
        if (t == 0 && live) { iters_out[cw] = cap_flag() != 0 ? NAN_MARK : iters; success_out[cw] = ok ? 1 : 0; cap_flag() = 0; }
    }
    // next group: the rest of this chunk, then the claimed chunk (collected during this chunk's first decode, many
    // barriers ago) or the static stride
    ++g;
    fresh = false;
    if (g >= g_end || g >= n_groups) {
        if (dyn) chunk = GEO::WG == 64 ? next_chunk : (uint32_t)__builtin_amdgcn_readfirstlane(*next_word);
        else chunk += gridDim.x;
        if constexpr (REDO) chunk = next_marked(chunk);
        g = chunk < n_chunks ? chunk * CLAIM_K : n_groups;
        g_end = g + CLAIM_K;
        fresh = true;
    }
    }                             // persistent loop over codeword groups
}


// PF: stage the NEXT codeword's LLRs in LDS with asynchronous global->LDS loads while the current
// one is being decoded (costs n*sizeof(T) bytes of LDS).
// LEAN: register-lean variant for the high-degree rate-4/5 codes (39 edges per index): the u of
// the exchanged edges is re-read from LDS in the check phase instead of being kept in VGPRs and
// the LLRs are re-read from global memory (L2) every iteration; this brings TM5120 under 128
// VGPRs so that two workgroups share a CU.  LEAN == 2 ("in place"): additionally the variable thread
// overwrites each exchanged u with nv = va - u in its LDS slot and publishes only the sign word of the
// marginals, so that no array of marginals is needed -- f64 TM8192 then fits the LDS (152 KB).
// Waves per SIMD the register allocation must leave room for.  The lean variant exists to reach 4
// (two 512-thread workgroups per CU); TM2048 sits right at the 80-VGPR step between 6 and 5 waves
// (f32 73, i8/i16 81 VGPRs: three workgroups per CU instead of two, 35 vs 30 M codewords/s for i8).
template <int CODE, class T, int IPT, int LEAN>
constexpr int min_waves_per_simd()
{
    // f64 on TC128 / TC256: 252-253 registers in round 2, 264-268 with round 3's additions -- one wave per SIMD instead of
    // two (694 -> 487, 383 -> 252 M codewords/s); held at 256
    if (sizeof(T) > 4) return (CODE <= TC256 && IPT == 1 && LEAN == 0) ? 2 : 1;
    // TC512 / TC128 f32 with their LLRs in LDS (llr_in_lds(): TC512 138 -> 128 registers, four values spilled around the loops, none
    // inside): four waves per SIMD WITH the wave verdict: TC512 +2.5 % at 2 dB (config 2), +4.5 % at 3 dB, +5.5 % at 5 dB; TC128 +10 % at
    // 2 and 3 dB (profiles/r06_kbench/tc512_llr_lds.txt)
    if (llr_in_lds<CODE, T, IPT, LEAN>()) return 4;
    if (LEAN == 1) return 4;
    // i32 on TC512: 127 -> 129 registers with the queue plumbing (which one-wave workgroups never use): held at 128
    if (CODE == TC512 && IPT == 1 && std::is_same_v<T, int32_t>) return 4;
    if (CODE == TM2048 && IPT == 1) return LDPC_TM2048_WAVES;
    if (CODE == TM1280 && IPT == 1) return 3;         // 183 -> 168 VGPRs: f32 28.4 -> 38.6, i8 27.0 -> 34.4 M codewords/s
    // TC codes (one-wave workgroups, occupancy set by registers alone): at 139-149 VGPRs three waves per SIMD.  Capping
    // them at 128 for four waves paid before the wave verdict existed (TC128 f32 507 -> 561, TC256 365 -> 391 M
    // codewords/s at 3 dB); the verdict's early exit costs 9-13 registers, which at 128 spill inside the loop
    // (TC512 f32 273 -> 180), while at three waves it is a net gain everywhere but TC128 at 3 dB (-4 %):
    //   f32, 3 dB / 5 dB, against four waves without verdict: TC128 534 / 1867 (556 / 1702), TC256 400 / 926 (394 / 783),
    //   TC512 277 / 543 (273 / 485); i8: TC128 502 / 1798 (522 / 1561), TC256 383 / 948 (351 / 789), TC512 285 / 561 (255 / 462).
    if (CODE <= TC512 && IPT == 1) return LDPC_WAVE_VERDICT ? 1 : (!(CODE >= TC256 && sizeof(T) < 4) ? 4 : 1);
    return 1;
}

template <int CODE, class T, int IPT, bool PF, int LEAN, int FORM, int NANPASS>
__device__ __forceinline__ void decode_ms_kernel_main(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                                                      uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                                                      uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim, uint32_t claim_k)
{
    using GEO = Geometry<CODE, T, IPT>;
    constexpr int Q = GEO::M / 4;
    constexpr int ESZ = (int)sizeof(typename Ops<T>::E);
    constexpr int GROUP_BYTES = ((LEAN == 2 ? GEO::NX * GEO::M * ESZ + GEO::NXC * GEO::M * 4 : (GEO::NX + GEO::NXC) * GEO::M * ESZ) + 8 + 15) / 16 * 16
                                + (llr_in_lds<CODE, T, IPT, LEAN>() ? (CODES[CODE].n / GEO::M) * GEO::M * ESZ : 0);
    __shared__ __attribute__((aligned(16))) char lds[GEO::G * GROUP_BYTES];
    __shared__ __attribute__((aligned(16))) char stage[PF ? CODES[CODE].n * sizeof(T) : 16];
    // Waves of a workgroup whose threads own two quarters' worth of indices (TM8192: 1024 threads,
    // quarters of 512) differ only in which quarter they start in: one wave-uniform branch
    // selects a body with all pi_k constants folded (the barriers inside both copies count
    // arrivals of the whole workgroup, whichever copy a wave runs).
    if constexpr (LDPC_QUARTER_SPECIALISE && GEO::G == 1 && GEO::NT == 2 * Q && Q >= 64) {
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) < Q)
            decode_ms_body<CODE, T, IPT, PF, LEAN, 0, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
        else
            decode_ms_body<CODE, T, IPT, PF, LEAN, 1, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
    } else if constexpr (LDPC_QUARTER_SPECIALISE >= 2 && GEO::G == 1 && GEO::NT == 4 * Q && Q >= 64) {
        const int jw = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / Q;
        if (jw == 0) decode_ms_body<CODE, T, IPT, PF, LEAN, 0, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
        else if (jw == 1) decode_ms_body<CODE, T, IPT, PF, LEAN, 1, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
        else if (jw == 2) decode_ms_body<CODE, T, IPT, PF, LEAN, 2, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
        else decode_ms_body<CODE, T, IPT, PF, LEAN, 3, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
    } else {
        decode_ms_body<CODE, T, IPT, PF, LEAN, -1, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k, lds, stage);
    }
}

template <int CODE, class T, int IPT, bool PF, int LEAN, int FORM = selfcorr_med3<CODE, T>(), int NANPASS = 0>
__global__ void __launch_bounds__((Geometry<CODE, T, IPT>::WG), (min_waves_per_simd<CODE, T, IPT, LEAN>()))
decode_ms_kernel(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                 uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                 uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim, uint32_t claim_k)
{
    decode_ms_kernel_main<CODE, T, IPT, PF, LEAN, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k);
}

// The same kernel for the ONE-WORKGROUP launches of single-frame host calls (notify.hpp): two more arguments and a completion ticket
// stored at the end.  A kernel of its own, not an argument of decode_ms_kernel: the two arguments alone cost the metric kernel 0.9 %
// (profiles/r06_kbench/launch_floor.txt); built for the codes of up to 2 048 bits, where a synchronisation is a fifth of a call.
template <int CODE> constexpr bool notify_kernel_built() { return CODES[CODE].n <= 2048; }

template <int CODE, class T, int IPT, bool PF, int LEAN, int FORM = selfcorr_med3<CODE, T>(), int NANPASS = 0>
__global__ void __launch_bounds__((Geometry<CODE, T, IPT>::WG), (min_waves_per_simd<CODE, T, IPT, LEAN>()))
decode_ms_notify_kernel(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                        uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                        uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim, uint32_t claim_k,
                        uint32_t *notify, uint32_t notify_ticket)
{
    decode_ms_kernel_main<CODE, T, IPT, PF, LEAN, FORM, NANPASS>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, claim_k);
    notify_done(notify, notify_ticket);
}

}  // namespace ldpc
