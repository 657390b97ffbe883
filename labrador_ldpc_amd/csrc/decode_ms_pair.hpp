// decode_ms_pair.hpp -- min-sum decoder (decode_ms<T>, /root/reference/src/decoder.rs:347-475) with
// PAIR ownership: thread t owns the ADJACENT indices 2t and 2t+1 (check 2t, 2t+1 of every block row
// and variable 2t, 2t+1 of every block column) instead of t and t + M/2 as decode_ms_kernel.hpp does.
//
// Why: the LDS, not the VALU, bounds the variable phase of the big codes, and a
// ds_write_b32 costs twice a ds_read_b32.  With adjacent indices
//   * every access at the thread's own position (u reads and marginal stores of the variable phase)
//     is one 64-bit LDS operation for both indices;
//   * a pi_k block rotates a quarter by phi: the images of 2t and 2t+1 are adjacent, and when phi is
//     EVEN they form an aligned pair -- one ds_read_b64 / ds_write_b64 and ONE address for both edges;
//     for odd phi the two edges keep their 32-bit accesses (20 of the 32 (block, quarter) rotations of
//     TM8192 are even);
//   * both indices of a thread lie in the same quarter, so every rotation constant is a literal after
//     one wave-uniform branch on the quarter (four copies of the body).
// Everything else -- arithmetic, order of accumulation, exclusive minima, sign words, flags, persistent
// workgroups, wave priorities -- is decode_ms_kernel.hpp's, whose helpers are used here.  Results are
// identical bit for bit (tests/test_gpu_parity.py runs this kernel as a `variant`).
//
// Preconditions (static_asserts): 4-byte register/LDS element types (f32, i8, i16), every exchanged
// block a pi_k, M/8 a multiple of 64 (the 64 lanes of a wave = 128 indices stay inside one quarter).
#pragma once

#include "decode_ms_kernel.hpp"

// Tuned settings (LDPC_PAIR_*, LDPC_PRIO_ROWS_PAIR): decode_ms_tuning.hpp.

namespace ldpc {


typedef float ldpc_f2 __attribute__((ext_vector_type(2)));

template <int CODE, class T>
struct PairGeometry {
    static constexpr int M = CODES[CODE].m;
    static constexpr int NT = M / 2;                         // threads per codeword = workgroup size
    static constexpr int NX = count_exchanged(*CODES[CODE].proto);
    static constexpr int NXC = count_exch_cols(*CODES[CODE].proto);
    static constexpr int OUT_LEN = CODES[CODE].output_len();
    static constexpr int LDS_BYTES = ((NX + NXC) * M * 4 + 16 + 15) / 16 * 16;      // exchange slots + two flags, the clamp vote, the next-codeword word
    static_assert(M % 512 == 0 && NT <= 1024, "pair ownership needs M/8 >= 64 lanes per quarter and <= 1024 threads");
};

// rank of local edge (S, B) among a thread's local edges, index-major
constexpr int pair_local_rank(const Prototype &p, int S, int B)
{
    int nloc = 0, r = 0;
    for (int b = 0; b < p.n_blocks; ++b)
        if (blk_local(p.blk[b])) { if (b < B) ++r; ++nloc; }
    return S * nloc + r;
}

constexpr bool all_exchanged_are_pi(const Prototype &p)
{
    for (int b = 0; b < p.n_blocks; ++b)
        if (!blk_local(p.blk[b]) && p.blk[b].kind != BLK_P) return false;
    return true;
}

// FORM: the self-correction's form (Ops::self_correct): 0 = compare + select, 2 / 3 = the clamp forms
template <int CODE, class T, int JW, int FORM>
LDPC_DEV void decode_ms_pair_body(const T *__restrict__ llrs, uint8_t *__restrict__ output,
                                  uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                                  uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim, char *lds)
{
    using GEO = PairGeometry<CODE, T>;
    using O = Ops<T>;
    using R = typename O::R;
    static_assert(sizeof(R) == 4 && sizeof(typename O::E) == 4, "pair kernel: 4-byte message types");
    constexpr Prototype P = *CODES[CODE].proto;
    static_assert(all_exchanged_are_pi(P), "pair kernel: exchanged blocks must be pi_k");
    constexpr int M = GEO::M, NT = GEO::NT, NB = P.n_blocks, NROWS = P.n_rows, NCOLS = P.n_cols;
    constexpr int N = CODES[CODE].n, NTX = N / M, NX = GEO::NX, NXC = GEO::NXC;
    constexpr int Q = M / 4, IPT = 2;
    constexpr int BLK_BYTES = M * 4, FLAG_OFF = (NX + NXC) * BLK_BYTES;
    // odd rotations read their two marginals as halves of two aligned 64-bit pairs (see check_phase): +12 % for
    // i8 (8.0 -> 9.0 M codewords/s).  For f32 the wider destinations cost nine spilled VGPRs in round 1 (6.7 -> 6.5) and
    // nothing either way in round 2 (7.00 / 7.00); with round 3's shorter check phase (the clamp form of the self-correction
    // frees the VCC round trips and two registers) the 20 % of LDS cycles lost to the 2-way conflicts of the 32-bit reads
    // show: 8.32-8.33 -> 8.43-8.45 M codewords/s, 2 spilled registers instead of 3 (profiles/r03_kbench/kb13.txt).
    constexpr bool ODD_B64 = LDPC_PAIR_ODD_B64 >= 0 ? LDPC_PAIR_ODD_B64 != 0 : true;
    constexpr int LOCAL_IN_VAR = LDPC_PAIR_LOCAL_IN_VAR >= 0 ? LDPC_PAIR_LOCAL_IN_VAR : (std::is_same_v<T, float> ? 10 : 8);
    constexpr bool PRIO_WAVES = true;            // LDPC_SETPRIO (decode_ms_kernel.hpp)
    (void)PRIO_WAVES;

    const int t = (int)(threadIdx.x & (M / 8 - 1)) + JW * (M / 8);
    __builtin_assume(t >= 0 && t < NT);
    const uint32_t n_groups = batch;
    uint32_t cw = blockIdx.x;
    bool live = cw < batch;
    (void)live;

    auto lds1 = [&](int off) LDPC_INLINE -> float & { return *reinterpret_cast<float *>(lds + off); };
    auto lds2 = [&](int off) LDPC_INLINE -> ldpc_f2 & { return *reinterpret_cast<ldpc_f2 *>(lds + off); };
    auto flag_at = [&](uint32_t which) LDPC_INLINE -> int & { return *reinterpret_cast<int *>(lds + FLAG_OFF + 4 * (which & 1)); };
    auto cap_flag = [&]() LDPC_INLINE -> int & { return *reinterpret_cast<int *>(lds + FLAG_OFF + 8); };
    constexpr bool NOCAP_POSSIBLE = LDPC_PAIR_NOCAP && std::is_same_v<T, float>;

    // Rotation of block B for this body's quarter JW: phi, and where the even/odd split puts the edges.
    // Byte address (inside the block region, biased as in decode_ms_kernel.hpp) of the variable that check
    // 2t + S is wired to; tb8 = 8 * t.  For even phi the address of S = 1 is that of S = 0 plus 4.
    auto even_c = [](int B) constexpr { return (phi_of(P.blk[B].val, JW, M) & 1) == 0; };
    auto wire = [&](auto B_, auto S_, int tb8) LDPC_INLINE -> int {
        constexpr int B = decltype(B_)::value, S = decltype(S_)::value;
        constexpr int K = P.blk[B].val;
        constexpr int base = (((theta_of(K) + JW) & 3) * Q) * 4 + lds_bias(P, B, BLK_BYTES);
        return ((tb8 + (phi_of(K, JW, M) + S) * 4) & (Q * 4 - 1)) | base;
    };

    // ---- state ---------------------------------------------------------------------------------------
    R u[IPT][NB], v[IPT][NB], va[IPT][NCOLS], llr[IPT][NTX];
    T lraw[IPT][NTX];

    auto fetch_llrs = [&](uint32_t c) LDPC_INLINE {
        unsigned tu = (unsigned)t;
        asm volatile("" : "+v"(tu));
        const uint32_t cc = c < batch ? c : batch - 1;
        static_for<0, NTX>([&](auto C_) LDPC_INLINE {
            constexpr int C = decltype(C_)::value;
            const T *src = (llrs + (size_t)cc * N) + (unsigned)(C * M);
            lraw[0][C] = src[2 * tu];
            lraw[1][C] = src[2 * tu + 1];
        });
    };

    auto begin_codeword = [&]() LDPC_INLINE {
        int tq = t;
        asm volatile("" : "+v"(tq));             // opaque per phase (no address hoisting), but visibly a multiple of 8 below
        const int tb8 = tq * 8;
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            constexpr int S = decltype(S_)::value;
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int B = decltype(B_)::value;
                u[S][B] = O::zero();                                                   // decoder.rs:374
                v[S][B] = O::zero();
            });
            static_for<0, NCOLS>([&](auto C_) LDPC_INLINE { va[S][decltype(C_)::value] = O::zero(); });
            static_for<0, NTX>([&](auto C_) LDPC_INLINE { llr[S][decltype(C_)::value] = O::load(lraw[S][decltype(C_)::value]); });
        });
        if constexpr (!LDPC_PAIR_PEEL_FIRST)
        static_for<0, NX>([&](auto X_) LDPC_INLINE {                                   // u = 0 in every exchange slot
            lds2(lds_xu_off(P, decltype(X_)::value, BLK_BYTES) + tb8) = ldpc_f2{0.0f, 0.0f};
        });
        if (t < 2) flag_at(t) = 0;
        // f32: the clamp of the exclusive minimum at FLT_MAX (decoder.rs:414-415) can only bite if some
        // magnitude reaches FLT_MAX, i.e. if an LLR is infinite or so large that sums overflow.  With every
        // |LLR| <= nocap_limit (derived from max_iters by the host: nocap_limit_for(), decode_ms_launch.hpp)
        // nothing can, and the check phase runs without the clamp operations (8 of 76 min-class
        // instructions per thread and iteration).
        if constexpr (NOCAP_POSSIBLE) {
            bool big = false;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                static_for<0, NTX>([&](auto C_) LDPC_INLINE {
                    const R a = O::mag(llr[decltype(S_)::value][decltype(C_)::value]);
                    big |= !(a <= nocap_limit) || (a != 0.0f && a < 0x1p-20f);     // NaN counts as out of range
                });
            });
            if (__ballot(big) != 0 && (t & 63) == 0) cap_flag() = 1;
        }
    };

    // BND_: the codeword passed the LLR range vote (it runs the clamp-free copy of the loop), which also makes
    // the multiply form of the self-correction test exact (Ops<float>::self_correct_b)
    auto edge_update = [&](auto S_, auto B_, R x, auto BND_) LDPC_INLINE {
        constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
        const R nv = O::sub_nv(x, u[S][B]);                                            // :421
        // the clamp forms need a guarantee about the values: integer messages always have it, f32 only the codewords of the
        // clamp-free loop (they passed the range vote)
        constexpr bool BND = decltype(BND_)::value != 0;
        constexpr int F = (FORM >= 2 && (BND || sizeof(T) <= 2)) ? FORM : (LDPC_PAIR_SELFCORR_CARRY != 0 ? 1 : 0);
        v[S][B] = O::template self_correct_b<BND, F>(nv, v[S][B]);                     // :422-425
    };

    // ---- variable phase: marginals (decoder.rs:382-383, :408) -------------------------------------
    // FIRST_: iteration 0 peeled (LDPC_PAIR_PEEL_FIRST): u == 0 and v == 0, see decode_ms_kernel.hpp PEEL_FIRST
    auto variable_phase = [&](auto BND_, auto FIRST_) LDPC_INLINE {
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        int tq = t;
        asm volatile("" : "+v"(tq));             // opaque per phase (no address hoisting), but visibly a multiple of 8 below
        const int tb8 = tq * 8;
        LDPC_SETPRIO(LDPC_PRIO_VAR);
        static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
            constexpr int C = decltype(C_)::value;
            if constexpr (C == NCOLS / 2) LDPC_SETPRIO(0);
            R acc0 = O::zero(), acc1 = O::zero();
            if constexpr (C < NTX) { acc0 = llr[0][C]; acc1 = llr[1][C]; }
            if constexpr (!FIRST)
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int B = decltype(B_)::value;
                if constexpr (P.blk[B].col == C) {
                    constexpr int slot = exch_slot(P, B);
                    if constexpr (slot >= 0) {
                        const ldpc_f2 up = lds2(lds_xu_off(P, slot, BLK_BYTES) + tb8);
                        acc0 = O::add(acc0, O::from_lds(up.x));
                        acc1 = O::add(acc1, O::from_lds(up.y));
                    } else {
                        acc0 = O::add(acc0, u[0][B]);
                        acc1 = O::add(acc1, u[1][B]);
                    }
                }
            });
            va[0][C] = acc0;
            va[1][C] = acc1;
            constexpr int cs = col_slot(P, C);
            if constexpr (cs >= 0) lds2(lds_xva_off(P, cs, BLK_BYTES) + tb8) = ldpc_f2{O::store(acc0), O::store(acc1)};
        });
        // the local-edge part of the check update for the first LDPC_PAIR_LOCAL_IN_VAR indices, while the
        // marginal stores drain (this phase is LDS-bound, the check phase VALU-bound)
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                if constexpr (exch_slot(P, B) < 0 && pair_local_rank(P, S, B) < LOCAL_IN_VAR) {
                    if constexpr (FIRST) v[S][B] = va[S][P.blk[B].col];
                    else edge_update(S_, B_, va[S][P.blk[B].col], BND_);
                }
            });
        });
    };

    // ---- check phase (decoder.rs:414-450 and :391-405 of the next iteration) -------------------------
    auto check_phase = [&](uint32_t it, auto CAP_, auto FIRST_) LDPC_INLINE {
        constexpr bool CAP = decltype(CAP_)::value != 0;
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        constexpr int BND = (!CAP && NOCAP_POSSIBLE) ? 1 : 0;
        int par_any = 0;
        int tq = t;
        asm volatile("" : "+v"(tq));             // opaque per phase (no address hoisting), but visibly a multiple of 8 below
        const int tb8 = tq * 8;
        LDPC_SETPRIO(3);
        R xs[IPT][NB];
        int ad[IPT][NB];
        auto request_marginals = [&]() LDPC_INLINE {
        static_for<0, NB>([&](auto B_) LDPC_INLINE {                                   // (1) request the exchanged marginals
            constexpr int B = decltype(B_)::value;
            constexpr int slot = exch_slot(P, B);
            if constexpr (slot >= 0) {
                constexpr int cs = col_slot(P, P.blk[B].col);
                constexpr int off = lds_xva_off(P, cs, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                if constexpr (even_c(B)) {
                    ad[0][B] = wire(B_, IC<0>{}, tb8);
                    const ldpc_f2 xp = lds2(off + ad[0][B]);
                    xs[0][B] = O::from_lds(xp.x);
                    xs[1][B] = O::from_lds(xp.y);
                } else if constexpr (ODD_B64) {
                    // odd phi: the two marginals are the HIGH half of one aligned pair and the LOW half of the next.
                    // Two ds_read_b64 of those pairs (2 LDS cycles each, conflict-free) instead of two
                    // ds_read_b32 whose 8-byte lane stride is a 2-way bank conflict (4 cycles each).
                    ad[0][B] = wire(B_, IC<-1>{}, tb8);                 // aligned pair (2t + phi - 1, 2t + phi)
                    ad[1][B] = wire(B_, IC<1>{}, tb8);                  // aligned pair (2t + phi + 1, 2t + phi + 2), wraps with the quarter
                    const ldpc_f2 lo = lds2(off + ad[0][B]), hi = lds2(off + ad[1][B]);
                    xs[0][B] = O::from_lds(lo.y);
                    xs[1][B] = O::from_lds(hi.x);
                } else {
                    ad[0][B] = wire(B_, IC<0>{}, tb8);
                    ad[1][B] = wire(B_, IC<1>{}, tb8);
                    xs[0][B] = O::from_lds(lds1(off + ad[0][B]));
                    xs[1][B] = O::from_lds(lds1(off + ad[1][B]));
                }
            }
        });
        };
        auto local_edges = [&]() LDPC_INLINE {
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {                                  // (2) local edges
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                if constexpr (exch_slot(P, B) < 0 && pair_local_rank(P, S, B) >= LOCAL_IN_VAR) {
                    if constexpr (FIRST) v[S][B] = va[S][P.blk[B].col];
                    else edge_update(S_, B_, va[S][P.blk[B].col], IC<BND>{});
                }
            });
        });
        };
        request_marginals();
        __builtin_amdgcn_sched_barrier(0);
        local_edges();
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, IPT>([&](auto S_) LDPC_INLINE {                                  // (3) exchanged edges
            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value, B = decltype(B_)::value;
                if constexpr (exch_slot(P, B) >= 0) {
                    if constexpr (FIRST) v[S][B] = xs[S][B];
                    else edge_update(S_, B_, xs[S][B], IC<BND>{});
                }
            });
        });
        static_for<0, NROWS>([&](auto R_) LDPC_INLINE {                                // (4) per check row, both indices
            constexpr int Rw = decltype(R_)::value;
            constexpr int D = row_degree(P, Rw);
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                constexpr int S = decltype(S_)::value;
                {   // (per-quarter tables against the age bias of the issue arbitration all measured <= this uniform one)
                    constexpr int prio_rows[6] = LDPC_PRIO_ROWS_PAIR;
                    constexpr int step = Rw * IPT + S, nsteps = IPT * NROWS;
                    constexpr int k = step - (nsteps - 6);
                    constexpr int now = k >= 0 ? prio_rows[k] : 3, before = (step > 0 && k >= 1) ? prio_rows[k - 1] : 3;
                    if constexpr (now != before) LDPC_SETPRIO(now);
                }
                R a[D], e[D];
                int sr[D], xw[D];
                static_for<0, D>([&](auto J_) LDPC_INLINE {
                    constexpr int J = decltype(J_)::value;
                    constexpr int B = row_block(P, Rw, J);
                    a[J] = v[S][B];
                    sr[J] = O::sign_word(v[S][B]);                                     // :439-441
                    if constexpr (exch_slot(P, B) >= 0) xw[J] = O::bits(xs[S][B]);     // :445-447
                    else xw[J] = O::bits(va[S][P.blk[B].col]);
                });
                const int sgn = xor_reduce<D>(sr), par = xor_reduce<D>(xw);
                exclusive_min<O, D, true, CAP>(a, e);                                  // :391-395, :430-435
                static_for<0, D>([&](auto J_) LDPC_INLINE {
                    constexpr int J = decltype(J_)::value;
                    constexpr int B = row_block(P, Rw, J);
                    u[S][B] = O::apply_sign(e[J], sgn, sr[J]);                         // :398-405
                    constexpr int slot = exch_slot(P, B);
                    if constexpr (slot >= 0) {
                        constexpr int off = lds_xu_off(P, slot, BLK_BYTES) - lds_bias(P, B, BLK_BYTES);
                        if constexpr (!even_c(B)) lds1(off + ad[S][B] + (ODD_B64 && S == 0 ? 4 : 0)) = O::store(u[S][B]);
                        else if constexpr (S == 1) lds2(off + ad[0][B]) = ldpc_f2{O::store(u[0][B]), O::store(u[1][B])};
                    }
                });
                par_any |= par;
            });
        });
        if (par_any < 0) flag_at(it) = 1;
    };

    // ---- persistent loop over codewords ----------------------------------------------------------------
    if (t == 0) cap_flag() = 0;
    LDPC_SYNC();
    // Where the next codeword's LLR loads are issued: before this one's epilogue (f32: 7.35 -> 7.66 M codewords/s, the
    // epilogue covers part of their latency) or at the top of its own turn (i8 / i16: the early loads' raw bytes are
    // spilled across the epilogue at the 128-register budget -- which also waits for them on the spot -- 48 spilled
    // registers against 3, 6.99 -> 7.09; fetching a column's two adjacent narrow LLRs with ONE load into ONE register halves
    // what the early fetch keeps alive, and still spills 23 registers against 6: 8.00 -> 7.71, profiles/r03_kbench/
    // kb22_pair_packed_fetch.txt).  LDPC_PAIR_FETCH_EARLY: -1 = per type, 0 / 1 = force.
    constexpr bool FETCH_EARLY = LDPC_PAIR_FETCH_EARLY >= 0 ? LDPC_PAIR_FETCH_EARLY != 0 : sizeof(T) >= 4;
    if (FETCH_EARLY && blockIdx.x < n_groups) fetch_llrs(blockIdx.x);
    // Dynamic distribution of the codewords (claim != nullptr): see decode_ms_body in decode_ms_kernel.hpp.  Thread 0
    // draws a ticket at the start of each decode; the codeword it names, gridDim.x + ticket, is this workgroup's NEXT
    // one.  The returning register is collected after the first pass (collect_claim) into one LDS word, which every wave
    // reads when the decode ends -- many barriers later -- in time for the early LLR fetch.
    const bool dyn = claim != nullptr && maxiters != 0;
    uint32_t ticket = 0;
    int *const next_word = reinterpret_cast<int *>(lds + FLAG_OFF + 12);
    auto collect_claim = [&]() LDPC_INLINE {
        if constexpr (JW == 0) {
            if (dyn && t == 0) {
                *next_word = (int)(gridDim.x + ticket);
                if (ticket == n_groups - 1) __hip_atomic_store(claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the launch's last draw
            }
        }
    };
    for (uint32_t g = blockIdx.x; g < n_groups;) {
        cw = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        if constexpr (JW == 0) {
            if (dyn && t == 0) ticket = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (!FETCH_EARLY) fetch_llrs(cw);   // (FETCH_EARLY: this codeword's LLR loads were issued behind the previous epilogue)
        begin_codeword();
        bool done = false, ok = false;
        uint32_t iters = maxiters;
        // the iterations, as one loop per clamp mode (two check phases inside ONE loop cost 330 spilled VGPRs)
        auto iterate = [&](auto CAP_) LDPC_INLINE {
            uint32_t it0 = 0;
            if constexpr (LDPC_PAIR_PEEL_FIRST != 0) {
                if (maxiters == 0) { done = true; return; }
                variable_phase(IC<(decltype(CAP_)::value == 0 && NOCAP_POSSIBLE) ? 1 : 0>{}, IC<1>{});
                LDPC_SYNC();
                check_phase(0u, CAP_, IC<1>{});
                collect_claim();
                it0 = 1;
            }
            for (uint32_t it = it0;; ++it) {
                if (it > 0) LDPC_SYNC();          // (the barrier before iteration 0 is taken below, before the clamp mode is read)
                if (it > 0 && flag_at(it - 1) == 0) { done = true; ok = true; iters = it - 1; }   // :453-463
                else if (it == maxiters) { done = true; }
                if (done) break;
                variable_phase(IC<(decltype(CAP_)::value == 0 && NOCAP_POSSIBLE) ? 1 : 0>{}, IC<0>{});
                LDPC_SYNC();
                if (it > 0 && t == 0) flag_at(it - 1) = 0;
                check_phase(it, CAP_, IC<0>{});
                if constexpr (LDPC_PAIR_PEEL_FIRST == 0) { if (it == 0) collect_claim(); }
            }
        };
        LDPC_SYNC();                              // the zeroed exchange slots, the flags and the clamp vote are visible
        if constexpr (NOCAP_POSSIBLE) {
            if (__builtin_amdgcn_readfirstlane(cap_flag()) != 0) iterate(IC<1>{});
            else iterate(IC<0>{});
        } else {
            iterate(IC<1>{});
        }
        // the next codeword's LLR loads, issued before this one's epilogue (fixed cost per codeword 2.55 -> 2.12 us)
        const uint32_t g_next = dyn ? (uint32_t)__builtin_amdgcn_readfirstlane(*next_word) : g + gridDim.x;
        if (FETCH_EARLY && g_next < n_groups) fetch_llrs(g_next);
        // hard decisions, MSB first (decoder.rs:455-461 / :467-473): lane l of a wave holds positions
        // 128w + 2l and 128w + 2l + 1, so the two ballots are interleaved bit by bit (scalar unit:
        // s_bitreplicate doubles every bit), then bit-reversed per byte
        static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
            constexpr int C = decltype(C_)::value;
            const unsigned long long ev = __ballot(O::bits(va[0][C]) < 0), od = __ballot(O::bits(va[1][C]) < 0);
            unsigned long long w[2];
            static_for<0, 2>([&](auto H_) LDPC_INLINE {
                constexpr int H = decltype(H_)::value;
                unsigned long long re, ro;
                const unsigned e32 = (unsigned)(ev >> (32 * H)), o32 = (unsigned)(od >> (32 * H));
                asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(re) : "s"(e32));
                asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(ro) : "s"(o32));
                const unsigned long long il = (re & 0x5555555555555555ull) | (ro & 0xAAAAAAAAAAAAAAAAull);   // bit p = position 64H + p
                const unsigned lo = __builtin_bswap32(__builtin_bitreverse32((unsigned)il));
                const unsigned hi = __builtin_bswap32(__builtin_bitreverse32((unsigned)(il >> 32)));
                w[H] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            });
            if ((t & 63) == 0) {
                unsigned long long *dst = reinterpret_cast<unsigned long long *>((output + (size_t)cw * GEO::OUT_LEN) + (C * M + 2 * t) / 8);
                dst[0] = w[0];
                dst[1] = w[1];
            }
        });
        if (t == 0) { iters_out[cw] = iters; success_out[cw] = ok ? 1 : 0; cap_flag() = 0; }
        LDPC_SYNC();
        g = g_next;
    }
}

// Self-correction form of the pair kernel.  The clamp forms (Ops<float>::clamp_to_side) replace the compare and the select by
// one v_med3_f32: TM8192 f32 7.56 -> 8.32 (form 2) / 8.07 (form 3) / 8.04 (form 5), i8 7.11 -> 7.78 M codewords/s (same-process
// A/B, profiles/r03_kbench/kb1.txt, kb10_forms.txt; identical outputs).  Form 2 is the faster one but narrows the f32 range vote
// (nocap_limit_for): the launcher uses it while that limit leaves room for real LLRs and form 3 beyond.
template <class T>
constexpr int pair_form_default()
{
    if (LDPC_PAIR_SELFCORR_MED3 >= 0) return LDPC_PAIR_SELFCORR_MED3;
    if (sizeof(T) <= 2) return 6;                // integer messages: two full-rate operations (IntOps::self_correct; i8 7.78 -> 7.99)
    return (sizeof(T) > 4 || std::is_same_v<T, int32_t>) ? 0 : 2;
}

template <int CODE, class T, int FORM = pair_form_default<T>()>
__global__ void __launch_bounds__((PairGeometry<CODE, T>::NT))
decode_ms_pair_kernel(const T *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters_out,
                      uint8_t *__restrict__ success_out, uint32_t batch, uint32_t maxiters, float nocap_limit, uint32_t *claim)
{
    using GEO = PairGeometry<CODE, T>;
    __shared__ __attribute__((aligned(16))) char lds[GEO::LDS_BYTES];
    const int jw = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (GEO::M / 8);          // quarter of this wave's indices
    if (jw == 0) decode_ms_pair_body<CODE, T, 0, FORM>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, lds);
    else if (jw == 1) decode_ms_pair_body<CODE, T, 1, FORM>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, lds);
    else if (jw == 2) decode_ms_pair_body<CODE, T, 2, FORM>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, lds);
    else decode_ms_pair_body<CODE, T, 3, FORM>(llrs, output, iters_out, success_out, batch, maxiters, nocap_limit, claim, lds);
}

}  // namespace ldpc
