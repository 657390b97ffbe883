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
//  * COMPRESSED CHECK STATE.  The reference keeps u[E] and v[E] (decoder.rs:375-376).
//    Here a check keeps {min1, min2, sign} and each edge keeps v; u is re-derived from
//    them exactly as decoder.rs:391-405 does.  Everything lives in VGPRs for the whole
//    decode; the LLRs are read from HBM once, hard bits written once.
//
//  * ORDER.  Marginals are accumulated per variable in the reference's edge order
//    restricted to that variable (LLR first, then blocks by (block row, term)), with the
//    same single IEEE / saturating operations (decoder.rs:408), so floating-point and
//    saturating-integer results agree bit for bit.  Min/second-min, sign and parity
//    accumulation are order-independent (decoder.rs:430-447).
//
// One iteration = phase A1 (check side: u for the exchanged edges -> LDS) | barrier |
// phase A2 (variable side: marginals, decoder.rs:382-411) | barrier | phase B (check side:
// new v with self-correction, mins, signs, parity, decoder.rs:419-450).  The "all parities
// satisfied" vote (decoder.rs:453) is an LDS flag read after the next barrier.
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "codes.hpp"

namespace ldpc {

// ---- compile-time loop ------------------------------------------------------------------
template <int N> struct IC { static constexpr int value = N; constexpr operator int() const { return N; } };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
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

// ---- arithmetic per LLR type: DecodeFrom, decoder.rs:22-86 ----------------------------------
template <class T> struct Ops;

template <> struct Ops<float> {                       // decoder.rs:69-77
    using R = float;                                   // register type
    static __device__ __forceinline__ R zero() { return 0.0f; }
    static __device__ __forceinline__ R maxval() { return FLT_MAX; }
    static __device__ __forceinline__ R load(float x) { return x; }
    static __device__ __forceinline__ float store(R x) { return x; }
    static __device__ __forceinline__ R abs(R x) { return __builtin_fabsf(x); }   // sign-bit mask, :73
    static __device__ __forceinline__ R add(R a, R b) { return a + b; }           // :74
    static __device__ __forceinline__ R sub(R a, R b) { return a - b; }           // :75
    static __device__ __forceinline__ R negate(R x) { return -x; }
    static __device__ __forceinline__ bool neg(R x) { return x < 0.0f; }          // hard_bit, :76
    static __device__ __forceinline__ bool is_zero(R x) { return x == 0.0f; }
    static __device__ __forceinline__ bool eq(R a, R b) { return a == b; }
    // two smallest of {m1, m2, a}: equals the strict-< update of decoder.rs:430-435
    static __device__ __forceinline__ void min2(R a, R &m1, R &m2)
    {
        m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
        m1 = __builtin_fminf(m1, a);
    }
};

template <class I, int LO, int HI> struct IntOps {     // decoder.rs:42-59
    using R = int;
    static __device__ __forceinline__ R zero() { return 0; }
    static __device__ __forceinline__ R maxval() { return HI; }
    static __device__ __forceinline__ R load(I x) { return (int)x; }
    static __device__ __forceinline__ I store(R x) { return (I)x; }
    static __device__ __forceinline__ R clamp(R x) { return x < LO ? LO : (x > HI ? HI : x); }
    static __device__ __forceinline__ R abs(R x) { R a = x < 0 ? -x : x; return a > HI ? HI : a; } // saturating_abs
    static __device__ __forceinline__ R add(R a, R b) { return clamp(a + b); }    // saturating_add
    static __device__ __forceinline__ R sub(R a, R b) { return clamp(a - b); }    // saturating_sub
    static __device__ __forceinline__ R negate(R x) { return -x; }
    static __device__ __forceinline__ bool neg(R x) { return x < 0; }
    static __device__ __forceinline__ bool is_zero(R x) { return x == 0; }
    static __device__ __forceinline__ bool eq(R a, R b) { return a == b; }
    static __device__ __forceinline__ void min2(R a, R &m1, R &m2)
    {
        const R lo = a < m1 ? a : m1, hi = a < m1 ? m1 : a;   // min/max(a, m1)
        m2 = hi < m2 ? hi : m2;
        m1 = lo;
    }
};
template <> struct Ops<int8_t>  : IntOps<int8_t, -128, 127> {};
template <> struct Ops<int16_t> : IntOps<int16_t, -32768, 32767> {};

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
    static constexpr size_t LDS_BYTES = (size_t)G * (NX + NXC) * M * sizeof(T) + G * 2 * sizeof(int);
    static_assert(M % IPT == 0 && NT >= 8 && (NT & (NT - 1)) == 0, "bad IPT");
};

// pi_k(i) for a run-time i; collapses to literals when the quarter index is known at compile time
template <int K, int M>
__device__ __forceinline__ int pi_dev(int i)
{
    constexpr int LQ = ilog2(M / 4), Q = M / 4;
    constexpr int P0 = phi_of(K, 0, M), P1 = phi_of(K, 1, M), P2 = phi_of(K, 2, M), P3 = phi_of(K, 3, M);
    constexpr int TH = theta_of(K);
    const int j = i >> LQ;
    const int phi = j == 0 ? P0 : (j == 1 ? P1 : (j == 2 ? P2 : P3));
    return (((TH + j) & 3) << LQ) + ((phi + i) & (Q - 1));
}

template <int CODE, class T, int IPT>
__global__ void __launch_bounds__((Geometry<CODE, T, IPT>::WG))
decode_ms_kernel(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                 uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                 uint32_t batch, uint32_t maxiters)
{
    using GEO = Geometry<CODE, T, IPT>;
    using O = Ops<T>;
    using R = typename O::R;
    constexpr Prototype P = *CODES[CODE].proto;
    constexpr int M = GEO::M, NT = GEO::NT, G = GEO::G, NB = GEO::NB, NROWS = GEO::NROWS,
                  NCOLS = GEO::NCOLS, NTX = GEO::NTX, NX = GEO::NX, NXC = GEO::NXC;
    constexpr int N = CODES[CODE].n;

    __shared__ T xu[G][NX * M];       // check -> variable messages of the exchanged blocks
    __shared__ T xva[G][NXC * M];     // marginals of the block columns those blocks touch
    __shared__ int unsat[G][2];       // "some parity check failed" per codeword, double-buffered

    const int tid = threadIdx.x;
    const int grp = G == 1 ? 0 : tid / NT;
    const int t = G == 1 ? tid : tid % NT;
    __builtin_assume(t >= 0 && t < NT);
    const uint32_t cw = blockIdx.x * G + grp;
    const bool live = cw < batch;

    // variable index (inside its block column) that check i of block B is wired to
    auto wire = [&](auto B, int i) -> int {
        constexpr Block blk = P.blk[B];
        if constexpr (blk.kind == BLK_I) return (i + blk.val) & (M - 1);
        else return pi_dev<blk.val, M>(i);
    };

    // ---- state, all in registers -----------------------------------------------------------
    R v[IPT][NB];                 // variable -> check message per edge        (decoder.rs:376)
    R m1[IPT][NROWS], m2[IPT][NROWS];   // two smallest |v| per check           (decoder.rs:378)
    bool sg[IPT][NROWS];          // product of signs per check                 (decoder.rs:367)
    R va[IPT][NCOLS];             // marginals                                  (decoder.rs:377)
    R llr[IPT][NTX];              // channel LLRs, read from HBM once

    static_for<0, IPT>([&](auto S) {
        static_for<0, NB>([&](auto B) { v[S][B] = O::zero(); });               // decoder.rs:374
        static_for<0, NROWS>([&](auto Rw) { m1[S][Rw] = O::zero(); m2[S][Rw] = O::zero(); sg[S][Rw] = false; });
        static_for<0, NCOLS>([&](auto C) { va[S][C] = O::zero(); });
        static_for<0, NTX>([&](auto C) {
            const int i = S * NT + t;
            llr[S][C] = live ? O::load(llrs[(size_t)cw * N + C * M + i]) : O::zero();
        });
    });
    if (t < 2) unsat[grp][t] = 0;
    __syncthreads();

    auto emit = [&](uint32_t iters, bool ok) {
        // hard decision of the marginals, MSB first (decoder.rs:455-461 / :467-473)
        static_for<0, IPT>([&](auto S) {
            static_for<0, NCOLS>([&](auto C) {
                const unsigned long long bits = __ballot(O::neg(va[S][C]));
                const int i = S * NT + t;
                if constexpr (NT >= 64) {
                    if ((tid & 63) == 0 && live) {
                        const unsigned long long w = __builtin_bswap64(__builtin_bitreverse64(bits));
                        *reinterpret_cast<unsigned long long *>(output + (size_t)cw * GEO::OUT_LEN + (C * M + i) / 8) = w;
                    }
                } else {
                    if ((tid & 7) == 0 && live) {
                        const unsigned b8 = (unsigned)(bits >> (tid & 63)) & 0xFFu;
                        output[(size_t)cw * GEO::OUT_LEN + (C * M + i) / 8] = (uint8_t)(__builtin_bitreverse32(b8) >> 24);
                    }
                }
            });
        });
        if (t == 0 && live) { iters_out[cw] = iters; success_out[cw] = ok ? 1 : 0; }
    };

    bool done = false;
    for (uint32_t it = 0;; ++it) {
        // ---- phase A1: check -> variable messages (decoder.rs:391-405) ------------------------
        R u[IPT][NB];
        static_for<0, IPT>([&](auto S) {
            const int i = S * NT + t;
            static_for<0, NB>([&](auto B) {
                constexpr Block blk = P.blk[B];
                constexpr int r = blk.row;
                const R mag = O::eq(O::abs(v[S][B]), m1[S][r]) ? m2[S][r] : m1[S][r];   // :391-395
                const bool flip = sg[S][r] != O::neg(v[S][B]);                           // :398-405
                u[S][B] = flip ? O::negate(mag) : mag;
                constexpr int slot = exch_slot(P, B);
                if constexpr (slot >= 0) xu[grp][slot * M + wire(B, i)] = O::store(u[S][B]);
            });
        });
        __syncthreads();

        // ---- verdict on the previous iteration (decoder.rs:453-463, :466-474) -----------------
        if (!done) {
            if (it > 0 && unsat[grp][(it - 1) & 1] == 0) { emit(it - 1, true); done = true; }
            else if (it == maxiters) { emit(maxiters, false); done = true; }
        }
        if constexpr (G == 1) { if (done) break; }
        else { if (__all(done)) break; }

        // ---- phase A2: marginals (decoder.rs:382-383, :408) -----------------------------------
        static_for<0, IPT>([&](auto S) {
            const int i = S * NT + t;
            static_for<0, NCOLS>([&](auto C) {
                R acc = O::zero();
                if constexpr (C < NTX) acc = llr[S][C];
                static_for<0, NB>([&](auto B) {
                    constexpr Block blk = P.blk[B];
                    if constexpr (blk.col == C) {
                        constexpr int slot = exch_slot(P, B);
                        if constexpr (slot >= 0) acc = O::add(acc, O::load(xu[grp][slot * M + i]));
                        else acc = O::add(acc, u[S][B]);
                    }
                });
                va[S][C] = acc;
                constexpr int cs = col_slot(P, C);
                if constexpr (cs >= 0) xva[grp][cs * M + i] = O::store(acc);
            });
        });
        __syncthreads();
        if (it > 0 && t == 0) unsat[grp][(it - 1) & 1] = 0;

        // ---- phase B: variable -> check messages (decoder.rs:414-450) -------------------------
        bool fail = false;
        static_for<0, IPT>([&](auto S) {
            const int i = S * NT + t;
            static_for<0, NROWS>([&](auto Rw) {
                R n1 = O::maxval(), n2 = O::maxval();                                    // :414-415
                bool sgn = false, par = false;                                           // :416-417
                static_for<0, NB>([&](auto B) {
                    constexpr Block blk = P.blk[B];
                    if constexpr (blk.row == Rw) {
                        constexpr int slot = exch_slot(P, B);
                        R x;
                        if constexpr (slot >= 0) x = O::load(xva[grp][col_slot(P, blk.col) * M + wire(B, i)]);
                        else x = va[S][blk.col];
                        const R nv = O::sub(x, u[S][B]);                                 // :421
                        const R old = v[S][B];
                        const bool keep = (O::neg(nv) == O::neg(old)) || O::is_zero(old); // :422
                        const R nw = keep ? nv : O::zero();                              // :423-425
                        v[S][B] = nw;
                        O::min2(O::abs(nw), n1, n2);                                     // :430-435
                        sgn ^= O::neg(nw);                                               // :439-441
                        par ^= O::neg(x);                                                // :445-447
                    }
                });
                m1[S][Rw] = n1; m2[S][Rw] = n2; sg[S][Rw] = sgn;
                fail |= par;
            });
        });
        if (fail) unsat[grp][it & 1] = 1;
    }
}

}  // namespace ldpc
