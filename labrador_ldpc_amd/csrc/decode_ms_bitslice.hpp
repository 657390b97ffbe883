// decode_ms_bitslice.hpp -- BIT-SLICED min-sum decoder for i8 LLRs on the CCSDS TM (AR4JA) codes.
//
// The same algorithm as decode_ms_kernel.hpp -- LDPCCode::decode_ms::<i8>, /root/reference/src/decoder.rs:347-475 with the
// i8 arithmetic of :42-50 -- in a different machine representation.  The f32-pipe kernels hold an i8 message in a 32-bit
// register (round 3's review: "the narrow types save only LLR bytes"); SDWA byte selects were measured at HALF rate on
// gfx950 (profiles/r04_kbench/sdwa_rate.txt), so packing bytes buys nothing.  Here a message costs what it carries, ONE
// BIT PER BIT: a 32-bit register holds ONE BIT PLANE of 32 different indices of a block, an i8 value is 8 registers, and
// every arithmetic step is a Boolean function on planes -- v_bitop3_b32 (any 3-input function, full rate) is the only
// arithmetic instruction of the iteration loop.  One wave instruction processes 64 lanes x 32 bits = 2048 edges.
//
//  * LAYOUT.  The TM codes' blocks are the identity or a CCSDS permutation pi_k, which moves quarter q of a block to quarter
//    (theta_k + q) mod 4 and rotates it by phi_k(q) (/root/reference/src/codes/mod.rs:313-317).  Index i = q * M/4 + j of a
//    block lives in lane (q, j mod L), bit j / L, with L = M/128 lanes per quarter.  A rotation of the quarter by phi is then
//    a lane permutation (phi mod L) plus ONE rotation of the 32-bit word (by phi / L, +1 where the lane index wraps):
//    ds_bpermute_b32 + v_alignbit_b32 per plane, no funnel of two words.  M/32 lanes hold a codeword (TM8192: the whole
//    wave; TM2048: 16 lanes, four codewords per wave, iterating in lockstep until the last one is done).
//
//  * STATE, compressed as hardware decoders keep it -- what DESIGN.md 7(a) costed as a loss for the f32 pipe is the natural
//    form here.  Per check row (per index bit): min1, min2 (7 planes each), the product of the signs, and the slot of the
//    edge that holds min1 (the reference's "|v| == min1 ? min2 : min1", decoder.rs:391-395, selects by VALUE; selecting by
//    that slot gives the same u: if two edges tie at min1 then min2 == min1).  Per edge: sign(v) and v != 0 -- all the
//    self-correction reads of the old v (decoder.rs:422).  TM8192: 90 planes = 90 VGPRs per lane hold the whole codeword.
//
//  * ORDER.  Block column by block column: marginal = LLR (+sat) u ... in the reference's edge order restricted to the
//    variable (decoder.rs:408), then at once the check side of the same edges: nv = va (-sat) u, self-correction, sign and
//    parity products, and the insertion of |v| into the row's two running minima (strict `<`, decoder.rs:430-434; as values
//    the two smallest of a multiset do not depend on the order).  The old row state is read-only during an iteration, the
//    new one replaces it at the end.  No barriers: a codeword never leaves its wave.
//
//  * |v| WITHOUT AN INCREMENT.  saturating_abs (decoder.rs:46) is v ^ sign + sign; the comparisons run on the key
//    2 * (v ^ sign) + sign instead, which orders like |v| with ties split by sign, and only the two minima of a row are
//    converted back, (key + 1) >> 1.  -128 takes the key of +127 (decoder.rs:46: |-128| = 127).
//
// The code below is written against a small backend `B` (one wave's registers as values of type B::V): HipBackend compiles it
// for gfx950, tests/bitslice_emu.cpp runs the same text lane by lane on the CPU, where it is compared with the oracle.
//
//  * SCHEDULE OF AN ITERATION (round 5).  The block columns are taken in an order that lets block row 0 -- three edges -- finish
//    first (Geo::COL_ORDER); a row's running state starts with its first edge (no all-ones initial state, no compare for that
//    edge) and REPLACES the row's old state the moment its last edge is done, so that old and new state of a row are live together
//    only between its first and its last edge.  Hard decisions live in LDS where that frees the registers for one more wave per
//    SIMD (Geo::HARD_LDS).
//
//  * WHAT AN INSTRUCTION COSTS (round 5, profiles/r05_kbench/permute_pipeline.txt).  At two waves per SIMD a wave is bound by ISSUE
//    -- one instruction of any kind per ~4.6 cycles, plus ~3.9 ns of SIMD time per ds_bpermute -- so the loop is written to issue
//    few: the next permutation job's eight ds_bpermute are in flight behind the current job's arithmetic, which lets ONE s_waitcnt
//    per job replace eight counted ones (Geo::PIPE); an exchanged edge's (sign, magnitude ^ sign) planes are kept from the variable
//    side for the check side (Geo::KEEP); the check side never forms the saturated va (-sat) u, it forms that value's KEY from the
//    wrapped difference and the overflow flag (Decoder::columns); a row's second edge needs one compare (min2 is still the largest
//    key); TM1280 rotates lanes with a DPP quad_perm instead of ds_bpermute (Geo::QUAD).
#pragma once

#include <cstdint>

#include "codes.hpp"

#ifndef BS_FN
#if defined(__HIPCC__)
#define BS_FN __device__ __forceinline__
#else
#define BS_FN inline
#endif
#endif

namespace ldpc {
namespace bs {

template <int N> struct IC { static constexpr int value = N; };
template <int B0, int E, class F>
BS_FN void sfor(F &&f)
{
    if constexpr (B0 < E) { f(IC<B0>{}); sfor<B0 + 1, E>(f); }
}

// truth tables of v_bitop3_b32: bit (a << 2 | b << 1 | c) of the constant is f(a, b, c); computed from the function itself
template <class F>
constexpr int tt_of(F f)
{
    int t = 0;
    for (int i = 0; i < 8; ++i) t |= (f((bool)((i >> 2) & 1), (bool)((i >> 1) & 1), (bool)(i & 1)) ? 1 : 0) << i;
    return t;
}
inline constexpr int TT_XOR3 = tt_of([](bool a, bool b, bool c) -> bool { return a ^ b ^ c; });
inline constexpr int TT_MAJ = tt_of([](bool a, bool b, bool c) -> bool { return (a && b) || (c && (a || b)); });
inline constexpr int TT_MUX = tt_of([](bool a, bool b, bool c) -> bool { return a ? b : c; });
inline constexpr int TT_OR3 = tt_of([](bool a, bool b, bool c) -> bool { return a || b || c; });
inline constexpr int TT_AND3 = tt_of([](bool a, bool b, bool c) -> bool { return a && b && c; });
// overflow of a + x in 8 bits from (a7, sign of x, sum7): same signs in, another sign out
inline constexpr int TT_OVF = tt_of([](bool a7, bool s, bool sum7) -> bool { return (a7 == s) && (sum7 != a7); });
// saturated plane 0..6: (ovf, a7, sum) -> ovf ? ~a7 : sum     (0x7F for a >= 0, 0x80 for a < 0)
inline constexpr int TT_SAT = tt_of([](bool ovf, bool a7, bool sum) -> bool { return ovf ? !a7 : sum; });
// The adder with its addend given INVERTED (x = ~xr, s = ~sr: subtracting what was added): the complements ride in the truth tables
inline constexpr int TT_XNOR3 = tt_of([](bool a, bool b, bool c) -> bool { return !(a ^ b ^ c); });                                   // a ^ ~b ^ c
inline constexpr int TT_MAJ_NB = tt_of([](bool a, bool b, bool c) -> bool { return (a && !b) || (c && (a || !b)); });                 // maj(a, ~b, c)
inline constexpr int TT_MAJ_NBC = tt_of([](bool a, bool b, bool c) -> bool { return (a && !b) || (!c && (a || !b)); });               // maj(a, ~b, ~c)
inline constexpr int TT_OVF_NS = tt_of([](bool a7, bool s, bool sum7) -> bool { return (a7 == !s) && (sum7 != a7); });               // overflow with sign ~s
// borrow of a - b, one plane: (b, a, borrow in) -> maj(~a, b, borrow)
inline constexpr int TT_BORROW = tt_of([](bool b, bool a, bool br) -> bool { return (!a && b) || (br && (!a || b)); });
// self-correction: (nz, sv, nv7) -> the old v was non-zero and the new sign differs
inline constexpr int TT_DROP = tt_of([](bool nz, bool sv, bool nv7) -> bool { return nz && (sv != nv7); });
// the same from the wrapped difference: (drop, ovf, sign) -> the value a forced plane takes, else the sign to XOR with
inline constexpr int TT_FVAL = tt_of([](bool drop, bool ovf, bool s) -> bool { return drop ? false : (ovf ? true : s); });
// (difference plane, force, fval) -> force ? fval : difference ^ fval
inline constexpr int TT_FORCE = tt_of([](bool d, bool force, bool fval) -> bool { return force ? fval : (d != fval); });
// plane 0 of the key: (sign, all1, key7) -> sign unless the value is -128
inline constexpr int TT_KEY0 = tt_of([](bool s, bool all1, bool k7) -> bool { return s && !(all1 && k7); });
static_assert(TT_XOR3 == 0x96 && TT_MAJ == 0xE8 && TT_MUX == 0xCA && TT_OR3 == 0xFE && TT_AND3 == 0x80);

constexpr int ilog2c(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// Bit planes of a message / marginal (two's complement).  8 = i8, what the library ships.  Everything below is written for PL planes: a
// build with -DBS_PLANES=16 decodes with i16 saturation (the experiment behind DESIGN.md section 7: sign-extended i8 LLRs through the
// i8 loader, checked against the i16 oracle in tests/test_bitslice_emu.py) -- twice the instructions and registers per edge.
#ifndef BS_PLANES
#define BS_PLANES 8
#endif
inline constexpr int PL = BS_PLANES, MG = PL - 1;          // planes of a value, of a magnitude
inline constexpr int LLRP = 8;                             // planes of an LLR as the (i8) loader delivers it
static_assert(PL == 8 || PL == 16);

// ---- geometry of a code in the bit-sliced layout ----------------------------------------------------------------------
// HALF = -1: one wave decodes a group of codewords alone.  HALF = 0 / 1: the wave is one of TWO that share a group, each owning a run of
// block columns with all their edges (the "split" kernel of the rate-4/5 codes, decode_ms_bitslice_split.hpp): everything below that
// counts edges, exchanged edges or transmitted columns then counts the OWNED ones.
template <int CODE, int HALF = -1>
struct Geo {
    static constexpr Prototype P = *CODES[CODE].proto;
    static constexpr int M = CODES[CODE].m, N = CODES[CODE].n, NP = N + CODES[CODE].p;
    static constexpr int Q = M / 4;                 // indices per quarter
    static constexpr int L = Q / 32;                // lanes per quarter
    static constexpr int W = 4 * L;                 // lanes per codeword
    static constexpr int G = 64 / W;                // codewords per wave
    static constexpr int NB = P.n_blocks, NROWS = P.n_rows, NCOLS = P.n_cols, NTX = N / M;
    static constexpr int OUT_LEN = NP / 8;
    static_assert(M >= 128 && (M & (M - 1)) == 0 && L >= 1 && W <= 64, "TM codes only");
    static constexpr int row_degree(int r) { int c = 0; for (int b = 0; b < NB; ++b) c += P.blk[b].row == r; return c; }
    static constexpr int max_row_degree() { int m = 0; for (int r = 0; r < NROWS; ++r) m = row_degree(r) > m ? row_degree(r) : m; return m; }
    static constexpr int ARG = ilog2c(max_row_degree());          // planes of the arg-min slot
    // slot of block b inside its row (the arg-min identifier)
    static constexpr int slot_of(int b) { int s = 0; for (int i = 0; i < b; ++i) s += P.blk[i].row == P.blk[b].row; return s; }
    static constexpr bool local(int b) { return P.blk[b].kind == BLK_I && P.blk[b].val == 0; }
    // the split: the set of block columns of half 0 (bit c of SPLIT_MASK) that balances the two waves' work best -- an exchanged edge
    // counted as 5/4 of a local one (its 16 ds_bpermute + v_alignbit), a block row with edges on both sides as the exchange and the
    // merge it costs each wave; found by trying every subset (at most 2^11)
    static constexpr int col_degree(int c) { int n = 0; for (int b = 0; b < NB; ++b) n += P.blk[b].col == c; return n; }
    static constexpr int half_cost(unsigned mask, int h)
    {
        int cost = 0;
        for (int b = 0; b < NB; ++b)
            if ((int)((mask >> P.blk[b].col) & 1u) == (h == 0 ? 1 : 0)) cost += local(b) ? 120 : 150;
        for (int r = 0; r < NROWS; ++r) {
            bool in0 = false, in1 = false;
            for (int b = 0; b < NB; ++b)
                if (P.blk[b].row == r) { if ((mask >> P.blk[b].col) & 1u) in0 = true; else in1 = true; }
            if (in0 && in1) cost += 200;
        }
        return cost;
    }
    static constexpr unsigned split_mask()
    {
        unsigned best = 1u;
        int best_c = 1 << 30;
        for (unsigned m = 1u; m + 1u < (1u << NCOLS); ++m) {
            if (!(m & 1u)) continue;                                       // (column 0 in half 0: one of each mirrored pair)
            const int c0 = half_cost(m, 0), c1 = half_cost(m, 1), c = c0 > c1 ? c0 : c1;
            if (c < best_c) { best_c = c; best = m; }
        }
        return best;
    }
    static constexpr unsigned SPLIT_MASK = split_mask();
    static constexpr bool in_half0(int c) { return (SPLIT_MASK >> c) & 1u; }
    static constexpr bool SPLIT = HALF >= 0;
    static constexpr bool owns_col(int c) { return HALF < 0 || in_half0(c) == (HALF == 0); }
    static constexpr bool owns(int b) { return owns_col(P.blk[b].col); }
    static constexpr bool row_in_half(int r, int h) { for (int b = 0; b < NB; ++b) if (P.blk[b].row == r && in_half0(P.blk[b].col) == (h == 0)) return true; return false; }
    static constexpr bool has_row(int r) { return HALF < 0 || row_in_half(r, HALF); }                      // this wave has edges in block row r
    static constexpr bool shared_row(int r) { return SPLIT && row_in_half(r, 0) && row_in_half(r, 1); }   // ... and so has the other
    // ordinal of block b among the OWNED exchanged (pi_k) blocks
    static constexpr int exch_of(int b) { int s = 0; for (int i = 0; i < b; ++i) s += (owns(i) && !local(i)) ? 1 : 0; return s; }
    static constexpr int NX = exch_of(NB);
    // ordinal of block column c among the owned transmitted ones, and their number
    static constexpr int llr_slot(int c) { int s = 0; for (int i = 0; i < c; ++i) s += (owns_col(i) && i < NTX) ? 1 : 0; return s; }
    static constexpr int NTX_OWN = llr_slot(NCOLS);
    // ordinal of block column c among the owned ones (hard-decision words)
    static constexpr int col_slot(int c) { int s = 0; for (int i = 0; i < c; ++i) s += owns_col(i) ? 1 : 0; return s; }
    static constexpr int NCOLS_OWN = col_slot(NCOLS);

    // ---- the order in which an iteration takes the owned block columns.  The columns that hold block row 0's edges come first (the
    // one with the fewest edges first): row 0 has three edges in two columns, so its running state is complete -- and becomes its old
    // state -- early in the iteration, and the other rows' running states start after it (they start with their first edge).
    struct ColOrder { int col[16]; int n; };
    static constexpr bool col_has_row(int c, int r) { for (int b = 0; b < NB; ++b) if (P.blk[b].col == c && P.blk[b].row == r) return true; return false; }
    static constexpr ColOrder col_order()
    {
        ColOrder o{};
        bool used[16] = {};
        for (int pass = 0; pass < 2; ++pass)                                     // pass 0: columns with an edge in block row 0
            for (;;) {
                int best = -1;
                for (int c = 0; c < NCOLS; ++c) {
                    if (used[c] || !owns_col(c) || (pass == 0 && !col_has_row(c, 0))) continue;
                    if (best < 0 || (pass == 0 && col_degree(c) < col_degree(best))) best = c;
                }
                if (best < 0) break;
                used[best] = true;
                o.col[o.n++] = best;
            }
        return o;
    }
    static constexpr ColOrder COL_ORDER = col_order();
    static constexpr int col_pos(int c) { for (int i = 0; i < COL_ORDER.n; ++i) if (COL_ORDER.col[i] == c) return i; return -1; }
    // position of owned edge e in the order its CHECK side is processed: column position, then block order inside the column
    static constexpr int proc_pos(int e) { return col_pos(P.blk[e].col) * MAX_BLOCKS + e; }
    // first / last owned edge of block row r in that order
    static constexpr int first_edge(int r)
    {
        int best = -1;
        for (int e = 0; e < NB; ++e) if (owns(e) && P.blk[e].row == r && (best < 0 || proc_pos(e) < proc_pos(best))) best = e;
        return best;
    }
    static constexpr int last_edge(int r)
    {
        int best = -1;
        for (int e = 0; e < NB; ++e) if (owns(e) && P.blk[e].row == r && (best < 0 || proc_pos(e) > proc_pos(best))) best = e;
        return best;
    }
    // the owned edge of block row r that follows first_edge(r) in that order (-1: the row has one owned edge)
    static constexpr int second_edge(int r)
    {
        int best = -1;
        for (int e = 0; e < NB; ++e)
            if (owns(e) && P.blk[e].row == r && e != first_edge(r) && (best < 0 || proc_pos(e) < proc_pos(best))) best = e;
        return best;
    }
    // the order in which an iteration uses the permutation-table entries (entry 2x = check -> variable alignment of exchanged edge x,
    // 2x + 1 = the way back): block column by block column in COL_ORDER, the variable side's edges, then the check side's
    struct PermOrder { int idx[2 * NB + 1]; int pos[2 * NB + 1]; };
    static constexpr PermOrder perm_order()
    {
        PermOrder o{};
        int n = 0;
        for (int i = 0; i < COL_ORDER.n; ++i)
            for (int side = 0; side < 2; ++side)
                for (int e = 0; e < NB; ++e)
                    if (P.blk[e].col == COL_ORDER.col[i] && owns(e) && !local(e)) o.idx[n++] = exch_of(e) * 2 + side;
        for (int i = 0; i < n; ++i) o.pos[o.idx[i]] = i;
        return o;
    }
    static constexpr PermOrder PERM_ORDER = perm_order();
    static constexpr int perm_after(int idx) { return PERM_ORDER.idx[(PERM_ORDER.pos[idx] + 1) % (2 * NX)]; }
    // The lane permutations of an iteration as a list of JOBS in that same order (job j uses table entry PERM_ORDER.idx[j]): side 0 =
    // an exchanged edge's u from check to variable alignment, side 1 = its block column's marginal the way back.  A job's eight
    // ds_bpermute can be issued one job AHEAD of their use (PIPE) when their source exists by then: u depends on the OLD row state
    // only (any time), the marginal is complete once the column's variable side is (so only behind another side-1 job of the column).
    struct Jobs { int edge[2 * NB + 1]; int side[2 * NB + 1]; int n; };
    static constexpr Jobs jobs()
    {
        Jobs o{};
        for (int i = 0; i < COL_ORDER.n; ++i)
            for (int side = 0; side < 2; ++side)
                for (int e = 0; e < NB; ++e)
                    if (P.blk[e].col == COL_ORDER.col[i] && owns(e) && !local(e)) { o.edge[o.n] = e; o.side[o.n] = side; ++o.n; }
        return o;
    }
    static constexpr Jobs JOBS = jobs();
    static constexpr int job_of(int e, int side) { for (int j = 0; j < JOBS.n; ++j) if (JOBS.edge[j] == e && JOBS.side[j] == side) return j; return -1; }
#ifndef BS_PIPE
#define BS_PIPE 3
#endif
    // bit 0: the two-wave kernels, bit 1: rate 2/3, bit 2: rate 1/2 (which has no registers for it: 21-23 spilled values, -3 %).
    // What it buys is not hidden latency -- the SIMD's other wave hid that already -- but ONE s_waitcnt per job instead of one per
    // plane: a job's results have long arrived when they are collected (profiles/r05_kbench/permute_pipeline.txt: +1.2 ... +1.9 %).
    // M = 128 (TM1280): a quarter is ONE lane and a codeword one quad of lanes, so the lane permutation of a pi_k block is a rotation of
    // the quad by theta_k -- a DPP quad_perm on a v_mov_b32 (no LDS, no latency; nothing at all for theta_k = 0) instead of a
    // ds_bpermute_b32, which at two waves per SIMD costs the SIMD ~4.4 x a VALU instruction (profiles/r04_kbench/sdwa_rate.txt).
#ifndef BS_QUAD
#define BS_QUAD 1
#endif
    static constexpr bool QUAD = W == 4 && BS_QUAD;
    static constexpr int quad_ctrl(int e, int side)
    {
        const int th = theta_of(P.blk[e].val);
        int c = 0;
        for (int i = 0; i < 4; ++i) c |= ((side == 0 ? i - th : i + th) & 3) << (2 * i);
        return c;
    }
    static constexpr bool PIPE = ((SPLIT ? 1 : P.n_blocks > 20 ? 2 : 4) & BS_PIPE) != 0 && !QUAD;
    // job j + 1 is issued before job j's results are used
    static constexpr bool issues_next(int j)
    {
        if (!PIPE || j < 0 || j + 1 >= JOBS.n) return false;
        return JOBS.side[j + 1] == 0 || (JOBS.side[j] == 1 && P.blk[JOBS.edge[j]].col == P.blk[JOBS.edge[j + 1]].col);
    }
    static constexpr bool issued_early(int j) { return j > 0 && issues_next(j - 1); }
    // An exchanged edge's u is needed twice, at check alignment: permuted into the marginal (variable side) and subtracted from the
    // permuted marginal (check side).  For the first KEEP exchanged edges of a block column the (sign, magnitude ^ sign) planes formed
    // for the variable side are KEPT for the check side instead of formed again (the seven XORs and whatever of the row-state multiplexer
    // the compiler did not carry over by itself; the registers exist: the kept planes die at the start of the check side, before its peak).
    // Measured +0.4 ... +2 % on every code (profiles/r05_kbench/permute_pipeline.txt).
#ifndef BS_KEEP_SPLIT
#define BS_KEEP_SPLIT 3
#endif
#ifndef BS_KEEP_R23
#define BS_KEEP_R23 3
#endif
#ifndef BS_KEEP_R12
#define BS_KEEP_R12 3
#endif
    static constexpr int KEEP = SPLIT ? BS_KEEP_SPLIT : P.n_blocks > 20 ? BS_KEEP_R23 : BS_KEEP_R12;
    static constexpr int keep_rank(int e) { int s = 0; for (int i = 0; i < e; ++i) s += (owns(i) && !local(i) && P.blk[i].col == P.blk[e].col) ? 1 : 0; return s; }
    static constexpr bool kept(int e) { return owns(e) && !local(e) && keep_rank(e) < KEEP; }
    static constexpr int LLR_WORDS = NTX_OWN * LLRP * 64;          // words of LLR planes per wave
    // the rate-4/5 codes (39 edges: 218 planes of state before any temporary) exist only in the two-waves-per-group form
    static constexpr bool TWO_WAVES = P.n_blocks > 30;
    // Every state update of the rate-2/3 codes is pinned to its place in the program (Decoder::pin_update): without that the
    // instruction selector's data-flow order keeps values whose next use is an iteration away live to the end of the block.  The
    // rate-1/2 codes fit their registers without the pins and run 6 % faster with the freedom (TM8192 16.9 against 15.9 M codewords/s);
    // the two-wave form of the rate-4/5 codes measured 36.8 (no pins) against 36.3 / 34.4 (profiles/r04_kbench/split_rate.txt).
#ifndef BS_PINNED_R12
#define BS_PINNED_R12 0
#endif
#ifndef BS_PINNED_R23
#define BS_PINNED_R23 1
#endif
#ifndef BS_PINNED_SPLIT
#define BS_PINNED_SPLIT 0
#endif
    // bit 0: the state updates, bit 1: the uses of the old row state in edge_u, bit 2: the magnitudes of a finished row
    static constexpr int PINNED = SPLIT ? BS_PINNED_SPLIT : P.n_blocks > 20 ? BS_PINNED_R23 : BS_PINNED_R12;
    // Hard decisions in LDS instead of one register per block column (one ds_write per column and iteration; a read as well where a
    // wave holds several codewords, whose finished ones keep their decisions): the rate-1/2 codes, which that brings under the 168
    // registers of three waves per SIMD.
#ifndef BS_HARD_LDS
#define BS_HARD_LDS 1
#endif
    static constexpr bool HARD_LDS = !SPLIT && P.n_blocks <= 20 && BS_HARD_LDS;
    static constexpr int ROW_NEW = 2 * PL + 2 + ARG;                           // planes of a row's running state: two keys, sign, parity, arg-min
    // LDS of a wave: lane permutations of the exchanged blocks [NX][2 directions][64] as 16-bit entries (source lane address | rotate
    // amount << 8; constant for the kernel's lifetime), hard-decision words [NCOLS][64] -- which double as the staging slab (STAGE_BYTES)
    // of the LLR transposition (prologue only) --, LLR planes [NTX][8][64].
    // Split mode: a wave's private area is [permutations | LLR planes]; the hard-decision words of the epilogue alias the LLR planes
    // (dead by then) and the staging slab is the wave's exchange buffer, which lies behind both private areas (SplitLayout).
    static constexpr int LDS_PERM = 0, LDS_PERM_END = LDS_PERM + NX * 2 * 64 * 2;
    static constexpr int HARD_BYTES = NCOLS_OWN * 256 > 2304 ? NCOLS_OWN * 256 : 2304;             // (>= STAGE_BYTES, a multiple of 256)
    // Rate 1/2 (three waves per SIMD, 168 registers): the "v != 0" plane of the first NZ_LDS edges lives in LDS -- read once and written
    // once per iteration, at the edge's check side -- in the part of the staging slab that the hard-decision words leave free during
    // the iterations: four planes the register allocator would otherwise keep in scratch (spilled values 11/14 -> 8/10; TM8192 i8
    // +2...+3.5 %, TM2048 +1.5 %).  More than the slab holds costs a wave per CU: 6 planes -3 % (profiles/r05_kbench/spill_leak.txt).
#ifndef BS_NZ_LDS
#define BS_NZ_LDS 4
#endif
    static constexpr int NZ_LDS = HARD_LDS ? BS_NZ_LDS : 0;
    static constexpr int NZ_IN_SLAB = (HARD_BYTES - NCOLS_OWN * 256) / 256 < NZ_LDS ? (HARD_BYTES - NCOLS_OWN * 256) / 256 : NZ_LDS;
    static constexpr int nz_slot(int e) { int s = 0; for (int i = 0; i < e; ++i) s += owns(i) ? 1 : 0; return s; }       // ordinal among the owned edges
    static constexpr bool nz_in_lds(int e) { return owns(e) && nz_slot(e) < NZ_LDS; }
    static constexpr int LDS_HARD = LDS_PERM_END, LDS_STAGE = LDS_HARD,
                         LDS_LLR = SPLIT ? LDS_PERM_END : LDS_HARD + HARD_BYTES,
                         LDS_PRIVATE = LDS_LLR + LLR_WORDS * 4,
                         LDS_NZ_TAIL = LDS_PRIVATE,
                         LDS_BYTES = LDS_PRIVATE + (NZ_LDS - NZ_IN_SLAB) * 256;
    static constexpr int nz_addr(int e) { return nz_slot(e) < NZ_IN_SLAB ? LDS_HARD + (NCOLS_OWN + nz_slot(e)) * 256 : LDS_NZ_TAIL + (nz_slot(e) - NZ_IN_SLAB) * 256; }
    static_assert(!SPLIT || NCOLS_OWN * 256 <= LLR_WORDS * 4, "the epilogue's hard-decision words alias the LLR planes");
    static_assert(NCOLS <= 16);
};

// ---- arithmetic on bit planes -------------------------------------------------------------------------------------------
template <class B>
struct Arith {
    using V = typename B::V;
    template <int TT> static BS_FN V op3(V a, V b, V c) { return B::template bitop3<TT>(a, b, c); }

    // acc (+sat) w, i8 saturation (decoder.rs:47-48), for two's-complement acc[8] and a sign-magnitude w = (sr, magnitude) handed over as
    // xr[k] = magnitude[k] ^ sr (one's complement if negative; the
    // + 1 rides on the carry-in sr).  INV: acc (-sat) w instead -- every bit of the addend inverted, which costs nothing: the truth
    // tables absorb the complements.  So one xr serves the variable side (add u) and the check side (subtract the same u) of an edge.
    // the same sum WITHOUT the saturation's selects: the wrapped sum planes and the overflow flag (the check side forms the key of the
    // saturated value from these directly: Decoder::columns)
    template <bool INV>
    static BS_FN void add_x_wrapped(const V (&acc)[PL], V sr, const V (&xr)[MG], V (&sum)[PL], V &ovf)
    {
        sum[0] = op3<TT_XOR3>(acc[0], xr[0], sr);
        V c = INV ? op3<TT_MAJ_NBC>(acc[0], xr[0], sr) : op3<TT_MAJ>(acc[0], xr[0], sr);
        sfor<1, MG>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            sum[k] = INV ? op3<TT_XNOR3>(acc[k], xr[k], c) : op3<TT_XOR3>(acc[k], xr[k], c);
            c = INV ? op3<TT_MAJ_NB>(acc[k], xr[k], c) : op3<TT_MAJ>(acc[k], xr[k], c);
        });
        sum[MG] = INV ? op3<TT_XNOR3>(acc[MG], sr, c) : op3<TT_XOR3>(acc[MG], sr, c);
        ovf = INV ? op3<TT_OVF_NS>(acc[MG], sr, sum[MG]) : op3<TT_OVF>(acc[MG], sr, sum[MG]);
    }
    template <bool INV>
    static BS_FN void sat_add_x(V (&acc)[PL], V sr, const V (&xr)[MG])
    {
        V sum[PL];
        // stage 0: carry-in = the addend's sign; a ^ ~x ^ ~s = a ^ x ^ s
        sum[0] = op3<TT_XOR3>(acc[0], xr[0], sr);
        V c = INV ? op3<TT_MAJ_NBC>(acc[0], xr[0], sr) : op3<TT_MAJ>(acc[0], xr[0], sr);
        sfor<1, MG>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            sum[k] = INV ? op3<TT_XNOR3>(acc[k], xr[k], c) : op3<TT_XOR3>(acc[k], xr[k], c);
            c = INV ? op3<TT_MAJ_NB>(acc[k], xr[k], c) : op3<TT_MAJ>(acc[k], xr[k], c);
        });
        sum[MG] = INV ? op3<TT_XNOR3>(acc[MG], sr, c) : op3<TT_XOR3>(acc[MG], sr, c);    // the addend's top bit is its sign (sign extension)
        const V ovf = INV ? op3<TT_OVF_NS>(acc[MG], sr, sum[MG]) : op3<TT_OVF>(acc[MG], sr, sum[MG]);
        const V a7 = acc[MG];
        sfor<0, MG>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            acc[k] = op3<TT_SAT>(ovf, a7, sum[k]);
        });
        acc[MG] = op3<TT_MUX>(ovf, a7, sum[MG]);
    }

    // AND / OR of planes k[LO] .. k[HI - 1] (and of `seed`), three inputs per instruction
    template <int LO, int HI> static BS_FN V and_planes(const V (&k)[PL])
    {
        V acc = k[LO];
        sfor<0, (HI - LO - 1) / 2>([&](auto I_) { constexpr int i = LO + 1 + 2 * decltype(I_)::value; acc = op3<TT_AND3>(acc, k[i], k[i + 1]); });
        if constexpr ((HI - LO - 1) % 2 == 1) acc = B::and_(acc, k[HI - 1]);
        return acc;
    }
    template <int LO, int HI> static BS_FN V or_planes(V seed, const V (&k)[PL])
    {
        V acc = seed;
        sfor<0, (HI - LO) / 2>([&](auto I_) { constexpr int i = LO + 2 * decltype(I_)::value; acc = op3<TT_OR3>(acc, k[i], k[i + 1]); });
        if constexpr ((HI - LO) % 2 == 1) acc = B::or_(acc, k[HI - 1]);
        return acc;
    }

    // a < b for PL-plane unsigned keys: the borrow out of a - b
    static BS_FN V less_than(const V (&a)[PL], const V (&b)[PL])
    {
        V br = B::andn(b[0], a[0]);                           // ~a0 & b0
        sfor<1, PL>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            br = op3<TT_BORROW>(b[k], a[k], br);
        });
        return br;
    }
};

// ---- the decoder of one group of G codewords, executed by one wave --------------------------------------------------------
template <int CODE, class B, int HALF = -1>
struct Decoder {
    using V = typename B::V;
    using GEO = Geo<CODE, HALF>;
    using A = Arith<B>;
    static constexpr int M = GEO::M, L = GEO::L, W = GEO::W, G = GEO::G, NB = GEO::NB, NROWS = GEO::NROWS, NCOLS = GEO::NCOLS, NTX = GEO::NTX;
    static constexpr int ARG = GEO::ARG;
    template <int TT> static BS_FN V op3(V a, V b, V c) { return B::template bitop3<TT>(a, b, c); }

    // lane constants
    V lane, q, ll, cwbase;
    // old row state (read-only between a row's first and last edge of an iteration) and per-edge bits
    V m1[NROWS][MG], m2[NROWS][MG], S[NROWS], arg[NROWS][ARG];
    V sv[NB], nz[NB];
    // running state of the rows whose edges are being processed: keys (8 planes), sign product, parity, arg-min slot
    V W1[NROWS][PL], W2[NROWS][PL], Sn[NROWS], Pn[NROWS], argn[NROWS][ARG];
    V hard[NCOLS];                      // (unused with Geo::HARD_LDS)
    V fail;                             // OR of the parities of the rows finished in this iteration (decoder.rs:453)

    BS_FN void init_lane(B &b)
    {
        lane = b.lane();
        ll = B::and_(lane, B::c(L - 1));
        q = B::and_(B::shr(lane, ilog2c(L)), B::c(3));
        cwbase = B::and_(lane, B::c(~(W - 1) & 63));
    }

    BS_FN void reinit_lane()
    {
        B::pin(lane);
        ll = B::and_(lane, B::c(L - 1));
        q = B::and_(B::shr(lane, ilog2c(L)), B::c(3));
        cwbase = B::and_(lane, B::c(~(W - 1) & 63));
    }

    BS_FN V hard_addr(int c) const { return B::add(B::shl(lane, 2), B::c(GEO::LDS_HARD + GEO::col_slot(c) * 256)); }

    BS_FN void reset_state(B &b)
    {
        // decoder.rs:374: the working area is zeroed, so before iteration 0 min1 = min2 = 0, every v = 0, every sign product +
        sfor<0, NROWS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            if constexpr (GEO::has_row(r)) {
                sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; m1[r][k] = B::c(0); m2[r][k] = B::c(0); });
                S[r] = B::c(0);
                sfor<0, ARG>([&](auto K_) { arg[r][decltype(K_)::value] = B::c(0); });
            }
        });
        sfor<0, NB>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            if constexpr (GEO::owns(e)) {
                sv[e] = B::c(0);
                if constexpr (GEO::nz_in_lds(e)) b.lds_write32(B::add(B::shl(lane, 2), B::c(GEO::nz_addr(e))), B::c(0));
                else nz[e] = B::c(0);
            }
        });
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (GEO::owns_col(c)) {
                if constexpr (GEO::HARD_LDS) b.lds_write32(hard_addr(c), B::c(0));        // (max_iters = 0: all-zero output, decoder.rs:466-473)
                else hard[c] = B::c(0);
            }
        });
    }

    // the same for the lanes of `fresh` only (slot refill, decode_refill below): they start a codeword, the other lanes keep theirs
    BS_FN void reset_state(B &b, uint64_t fresh)
    {
        auto z = [&](V &x) { x = B::select_lanes(fresh, B::c(0), x); };
        const V fp = b.plane_of(fresh);
        sfor<0, NROWS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            if constexpr (GEO::has_row(r)) {
                sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; z(m1[r][k]); z(m2[r][k]); });
                z(S[r]);
                sfor<0, ARG>([&](auto K_) { z(arg[r][decltype(K_)::value]); });
            }
        });
        sfor<0, NB>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            if constexpr (GEO::owns(e)) {
                z(sv[e]);
                if constexpr (GEO::nz_in_lds(e)) b.lds_write32_if(B::add(B::shl(lane, 2), B::c(GEO::nz_addr(e))), B::c(0), fp);
                else z(nz[e]);
            }
        });
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (GEO::owns_col(c)) {
                if constexpr (GEO::HARD_LDS) b.lds_write32_if(hard_addr(c), B::c(0), fp);
                else z(hard[c]);
            }
        });
    }

    // "this edge holds the row's min1": arg[r] == slot.  An AND over ARG literals (plane k, or its complement where bit k of SLOT is 0):
    // three literals in the first instruction, two more per further one -- the complements ride in the truth tables
    template <int R, int SLOT>
    BS_FN V is_arg() const
    {
        constexpr auto bit = [](int k) { return (SLOT >> k) & 1; };
        if constexpr (ARG == 0) return B::c(0xFFFFFFFFu);
        else if constexpr (ARG == 1) return bit(0) ? arg[R][0] : B::not_(arg[R][0]);
        else if constexpr (ARG == 2) {
            constexpr int tt = tt_of([](bool, bool x1, bool x0) -> bool { return x1 == (bool)((SLOT >> 1) & 1) && x0 == (bool)(SLOT & 1); });
            return op3<tt>(arg[R][1], arg[R][1], arg[R][0]);
        } else {
            constexpr int tt3 = tt_of([](bool x2, bool x1, bool x0) -> bool {
                return x2 == (bool)((SLOT >> 2) & 1) && x1 == (bool)((SLOT >> 1) & 1) && x0 == (bool)(SLOT & 1);
            });
            V r = op3<tt3>(arg[R][2], arg[R][1], arg[R][0]);
            sfor<0, (ARG - 3) / 2>([&](auto I_) {
                constexpr int k = 3 + 2 * decltype(I_)::value;
                constexpr int tt = tt_of([](bool acc, bool xb, bool xa) -> bool {
                    return acc && xb == (bool)((SLOT >> (3 + 2 * decltype(I_)::value + 1)) & 1) && xa == (bool)((SLOT >> (3 + 2 * decltype(I_)::value)) & 1);
                });
                r = op3<tt>(r, arg[R][k + 1], arg[R][k]);
            });
            if constexpr ((ARG - 3) % 2 == 1) r = bit(ARG - 1) ? B::and_(r, arg[R][ARG - 1]) : B::andn(r, arg[R][ARG - 1]);
            return r;
        }
    }

    // u of edge E at CHECK alignment: sign su, magnitude mg[7]  (decoder.rs:391-405 from the compressed row state)
    template <int E>
    BS_FN void edge_u(V &su, V (&mg)[MG])
    {
        constexpr int r = GEO::P.blk[E].row, slot = GEO::slot_of(E);
        // (everything below depends on the OLD row state only, which exists from the top of the iteration: without the pins the
        // instruction selector computes later edges' u early and keeps it)
        sfor<0, ARG>([&](auto K_) { pin_use(arg[r][decltype(K_)::value]); });
        pin_use(S[r]);
        const V sel = is_arg<r, slot>();
        sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; mg[k] = op3<TT_MUX>(sel, m2[r][k], m1[r][k]); });
        su = B::xor_(S[r], sv[E]);
    }

    // ---- lane permutations of a pi_k block -------------------------------------------------------------------------
    // packed phi tables of pi_k for this code's M: low (phi mod L) and high (phi / L) parts of the four quarters
    static constexpr uint32_t phi_lo_lit(int k) { uint32_t v = 0; for (int j = 0; j < 4; ++j) v |= (uint32_t)(phi_of(k, j, M) % L) << (4 * j); return v; }
    static constexpr uint32_t phi_hi_lit(int k) { uint32_t v = 0; for (int j = 0; j < 4; ++j) v |= (uint32_t)(phi_of(k, j, M) / L) << (5 * j); return v; }

    // pull by VARIABLE lanes from CHECK lanes (u, check -> variable): source lane byte address and rotate-right amount
    template <int K>
    BS_FN void perm_c2v(V &addr, V &amt) const
    {
        constexpr int TH = theta_of(K);
        const V sq = B::and_(B::sub(q, B::c(TH)), B::c(3));                       // source quarter of the check
        const V plo = B::bfe(B::c(phi_lo_lit(K)), B::shl(sq, 2), 4);              // phi(source quarter) mod L
        const V phi = B::bfe(B::c(phi_hi_lit(K)), B::add(B::shl(sq, 2), sq), 5);  // phi / L
        const V t = B::sub(ll, plo);
        const V neg = B::sar(t, 31);                                              // -1 where the lane index wraps
        const V sl = B::and_(t, B::c(L - 1));
        addr = B::shl(B::add(cwbase, B::add(B::shl(sq, ilog2c(L)), sl)), 2);
        amt = B::sub(neg, phi);                                                   // rotl by (phi + wrap) = rotr by -(phi + wrap)
    }
    // pull by CHECK lanes from VARIABLE lanes (marginals, variable -> check)
    template <int K>
    BS_FN void perm_v2c(V &addr, V &amt) const
    {
        constexpr int TH = theta_of(K);
        const V dq = B::and_(B::add(q, B::c(TH)), B::c(3));                       // quarter of the variable
        const V plo = B::bfe(B::c(phi_lo_lit(K)), B::shl(q, 2), 4);               // phi(own quarter)
        const V phi = B::bfe(B::c(phi_hi_lit(K)), B::add(B::shl(q, 2), q), 5);
        const V t = B::add(ll, plo);
        const V over = B::shr(t, ilog2c(L));                                      // 1 where the lane index wraps (t < 2L)
        const V sl = B::and_(t, B::c(L - 1));
        addr = B::shl(B::add(cwbase, B::add(B::shl(dq, ilog2c(L)), sl)), 2);
        amt = B::add(phi, over);                                                  // rotr by phi + wrap
    }

    static BS_FN void pin_update(V &x) { if constexpr (GEO::PINNED & 1) B::pin(x); }
    static BS_FN void pin_use(V &x) { if constexpr (GEO::PINNED & 2) B::pin(x); }
    static BS_FN void pin_row(V &x) { if constexpr (GEO::PINNED & 4) B::pin(x); }

    template <int IDX> BS_FN V perm_entry(B &b) const { return b.lds_read_u16(B::add(B::shl(lane, 1), B::c(GEO::LDS_PERM + IDX * 128))); }
    // The entries are used in a fixed order (Geo::PERM_ORDER), so each use hands out the entry read during the PREVIOUS use and starts
    // the read of the next one: an LDS round trip in front of every edge's eight ds_bpermute, with one other wave on the SIMD to hide
    // it behind, cost 10-12 % (profiles/r04_kbench/bs_diag.txt).  One register.
    V pnext;
    BS_FN void prime_perm(B &b) { pnext = perm_entry<GEO::PERM_ORDER.idx[0]>(b); }
    template <int IDX> BS_FN V take_perm(B &b)
    {
        const V a = pnext;
        pnext = perm_entry<GEO::perm_after(IDX)>(b);
        return a;
    }
    // ---- a permutation job's eight ds_bpermute, issued (possibly a job ahead: Geo::PIPE) and collected ----
    V pend[PL], pamt;
    V kx[GEO::KEEP > 0 ? GEO::KEEP : 1][MG], ksu[GEO::KEEP > 0 ? GEO::KEEP : 1];
    template <int J> BS_FN void issue_job(B &b, const V (&va)[PL])
    {
        constexpr int e = GEO::JOBS.edge[J], side = GEO::JOBS.side[J];
        const V addr = take_perm<GEO::exch_of(e) * 2 + side>(b);
        pamt = B::shr(addr, 8);
        auto move = [&](V x) {
            if constexpr (GEO::QUAD) return B::template quad_perm<GEO::quad_ctrl(e, side)>(x);
            else return b.bperm(addr, x);
        };
        if constexpr (side == 0) {
            V su, mg[MG];
            edge_u<e>(su, mg);
            if constexpr (GEO::kept(e)) {
                constexpr int kr = GEO::keep_rank(e);
                sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; kx[kr][k] = B::xor_(mg[k], su); pend[k] = move(kx[kr][k]); });
                ksu[kr] = su;
            } else {
                sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; pend[k] = move(mg[k]); });
            }
            pend[MG] = move(su);
        } else {
            sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; pend[k] = move(va[k]); });
        }
    }
    // the results of job J, rotated into place; then the NEXT job's permutes go out, in front of this job's arithmetic
    template <int J> BS_FN void collect_job(B &b, V (&got)[PL], const V (&va)[PL])
    {
        if constexpr (!GEO::issued_early(J)) issue_job<J>(b, va);
        const V amt = pamt;
        if constexpr (GEO::PIPE) { B::fence(); B::lds_wait(); B::fence(); }         // (one s_waitcnt lgkmcnt(0) for the eight results)
        sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; got[k] = B::rotr(pend[k], amt); });
        if constexpr (GEO::PIPE) B::fence();
        if constexpr (GEO::issues_next(J)) { issue_job<J + 1>(b, va); B::fence(); }
    }

    template <int IDX> BS_FN void put_perm_entry(B &b, V w) const { b.lds_write16(B::add(B::shl(lane, 1), B::c(GEO::LDS_PERM + IDX * 128)), w); }

    // The permutations depend on the lane only: computed once per kernel into LDS.  ds_bpermute_b32 reads bits 7:2 of its address
    // and v_alignbit_b32 bits 4:0 of its shift, so one 16-bit entry carries both: address | amount << 8.
    BS_FN void init_perm_tables(B &b) const
    {
        sfor<0, NB>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            if constexpr (GEO::owns(e) && !GEO::local(e)) {
                V addr, amt;
                perm_c2v<GEO::P.blk[e].val>(addr, amt);
                put_perm_entry<GEO::exch_of(e) * 2 + 0>(b, B::or_(addr, B::shl(B::and_(amt, B::c(31)), 8)));
                perm_v2c<GEO::P.blk[e].val>(addr, amt);
                put_perm_entry<GEO::exch_of(e) * 2 + 1>(b, B::or_(addr, B::shl(B::and_(amt, B::c(31)), 8)));
            }
        });
    }

    // ---- a row's running state becomes its old state; minima back to magnitudes: (key + 1) >> 1 = (key >> 1) + (key & 1) ----
    template <int R> BS_FN void finish_row()
    {
        auto to_mag = [&](const V (&key)[PL], V (&mag)[MG]) {
            V c = key[0];
            sfor<0, MG>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                mag[k] = B::xor_(key[k + 1], c);
                c = B::and_(key[k + 1], c);
            });
        };
        to_mag(W1[R], m1[R]);
        to_mag(W2[R], m2[R]);
        S[R] = Sn[R];
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; arg[R][k] = argn[R][k]; });
        // (the magnitudes are formed HERE: left to the instruction selector, their last XORs sink to their first use in the next
        // iteration and keys and carries -- 24 planes instead of 14 -- stay live until then)
        sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; pin_row(m1[R][k]); pin_row(m2[R][k]); });
        fail = B::or_(fail, Pn[R]);                                                   // non-zero bits = unsatisfied checks (:453)
    }

    // ---- one iteration (decoder.rs:380-450), block column by block column -------------------------------------------
    // `frozen`: the lanes of codewords that are finished (their hard decisions stay as they are), as a lane mask
    BS_FN V iteration(B &b, uint64_t frozen)
    {
        columns(b, frozen);
        return finish_iteration(b);
    }
    // the owned block columns in Geo::COL_ORDER: variable side, then check side, of each.  A row that only this wave has edges in is
    // finished with its last edge; a row shared with another wave (split mode) stays a partial running state for the exchange.
    BS_FN void columns(B &b, uint64_t frozen)
    {
        fail = B::c(0);
        sfor<0, GEO::COL_ORDER.n>([&](auto I_) {
            constexpr int c = GEO::COL_ORDER.col[decltype(I_)::value];
            B::fence();            // keep the compiler from hoisting the next column's loads over this one's arithmetic (register pressure)
            // ---- variable side: marginal of block column c (decoder.rs:382-383, :408) ----
            V va[PL];
            if constexpr (c < NTX) {
                const V at = B::shl(lane, 2);
                sfor<0, PL>([&](auto K_) {                       // (planes beyond the loader's: the sign plane again -- sign extension)
                    constexpr int k = decltype(K_)::value;
                    va[k] = b.lds_read32(B::add(at, B::c(GEO::LDS_LLR + (GEO::llr_slot(c) * LLRP + (k < LLRP ? k : LLRP - 1)) * 256)));
                });
            } else {
                sfor<0, PL>([&](auto K_) { va[decltype(K_)::value] = B::c(0); });
            }
            // (u of a LOCAL edge -- an unshifted identity block: check and variable alignment coincide -- is the same expression here
            // and on the check side below, where the adder takes it inverted: the compiler forms it once)
            sfor<0, NB>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                if constexpr (GEO::P.blk[e].col == c) {
                    V su, mg[MG], x[MG];
                    if constexpr (!GEO::local(e)) {
                        V got[PL];
                        collect_job<GEO::job_of(e, 0)>(b, got, va);
                        sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; mg[k] = got[k]; });
                        su = got[MG];
                    } else {
                        edge_u<e>(su, mg);
                    }
                    if constexpr (GEO::kept(e)) sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; x[k] = mg[k]; });      // (permuted as x)
                    else sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; x[k] = B::xor_(mg[k], su); });
                    A::template sat_add_x<false>(va, su, x);
                    B::fence();
                }
            });
            // hard decisions (decoder.rs:457-461): a wave that holds ONE codeword stops with it, so nothing is ever frozen there
            if constexpr (GEO::HARD_LDS) {
                if constexpr (G == 1) b.lds_write32(hard_addr(c), va[MG]);
                else b.lds_write32(hard_addr(c), B::select_lanes(frozen, b.lds_read32(hard_addr(c)), va[MG]));   // (branch-free: an EXEC-masked
            } else if constexpr (G == 1) hard[c] = va[MG];                                                        //  store would split the loop body)
            else hard[c] = B::select_lanes(frozen, hard[c], va[MG]);
            // ---- check side of the same edges (decoder.rs:419-447) ----
            sfor<0, NB>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                if constexpr (GEO::P.blk[e].col == c) {
                    constexpr int r = GEO::P.blk[e].row, slot = GEO::slot_of(e);
                    V nv[PL];
                    if constexpr (!GEO::local(e)) {
                        collect_job<GEO::job_of(e, 1)>(b, nv, va);
                    } else {
                        sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; nv[k] = va[k]; });
                    }
                    const V pbit = nv[MG];                                            // hard bit of the marginal (:445-447)
                    V su, mg[MG], x[MG];
                    if constexpr (GEO::kept(e)) {
                        constexpr int kr = GEO::keep_rank(e);
                        su = ksu[kr];
                        sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; x[k] = kx[kr][k]; });
                    } else {
                        edge_u<e>(su, mg);
                        sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; x[k] = B::xor_(mg[k], su); });
                    }
                    // new_v_ai = va (-sat) u (:421) is never formed either: the WRAPPED difference and its overflow flag are enough.  The
                    // saturated value's sign is the marginal's on overflow, the difference's otherwise; its magnitude planes ^ sign are all
                    // ones on overflow (+127, and -128 which takes +127's key).
                    V dsum[PL], ovf;
                    A::template add_x_wrapped<true>(nv, su, x, dsum, ovf);
                    const V nsign = op3<TT_MUX>(ovf, nv[MG], dsum[MG]);
                    // self-correction (:422-426): keep unless the old v was non-zero with the other sign
                    V nz_old;
                    if constexpr (GEO::nz_in_lds(e)) nz_old = b.lds_read32(B::add(B::shl(lane, 2), B::c(GEO::nz_addr(e))));
                    else nz_old = nz[e];
                    const V drop = op3<TT_DROP>(nz_old, sv[e], nsign);
                    // v = drop ? 0 : nv: its sign, its key and "v != 0".  key of |v|: planes 1..7 = v ^ sign, plane 0 = sign -- except for
                    // -128, which is +127's key.  A key plane is FORCED to 0 where v is dropped and to 1 where the difference overflowed:
                    // (force, value) select among {difference ^ sign, 0, 1} in one instruction per plane
                    const V force = B::or_(ovf, drop);
                    const V fval = op3<TT_FVAL>(drop, ovf, nsign);
                    V key[PL];
                    sfor<0, MG>([&](auto K_) { constexpr int k = decltype(K_)::value; key[k + 1] = op3<TT_FORCE>(dsum[k], force, fval); });
                    const V vs = B::andn(nsign, drop);                               // sign of the new v
                    const V all1 = A::template and_planes<1, MG>(key);
                    key[0] = op3<TT_KEY0>(vs, all1, key[MG]);
                    sv[e] = vs;
                    if constexpr (GEO::nz_in_lds(e)) b.lds_write32(B::add(B::shl(lane, 2), B::c(GEO::nz_addr(e))), A::template or_planes<1, PL>(vs, key));
                    else nz[e] = A::template or_planes<1, PL>(vs, key);
                    if constexpr (e == GEO::first_edge(r)) {
                        // the row's running state STARTS with this edge: every key is <= 254 = the key of maxval (decoder.rs:414-415), so
                        // after one insertion min1 = this key and min2 = 254 whatever the key -- no compare; the arg-min slot is this
                        // edge's (if the key IS 254 the reference's strict `<` would leave the slot alone, but then min1 = min2 and the
                        // slot selects between equal values)
                        sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; W1[r][k] = key[k]; W2[r][k] = B::c(k == 0 ? 0u : 0xFFFFFFFFu); });
                        Pn[r] = pbit;
                        Sn[r] = vs;
                        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; argn[r][k] = B::c(((slot >> k) & 1) ? 0xFFFFFFFFu : 0u); });
                    } else {
                        Pn[r] = B::xor_(Pn[r], pbit);                                // parity of the marginals' hard bits
                        Sn[r] = B::xor_(Sn[r], vs);                                  // product of the signs (:438-441)
                        // two running minima (:430-434)
                        const V lt1 = A::less_than(key, W1[r]);
                        if constexpr (e == GEO::second_edge(r)) {
                            // min2 is still 254, the largest key there is: "key < min2 ? key : min2" IS key (a key that is not
                            // smaller is 254 itself) -- the row's second edge needs one compare and two selects per plane
                            sfor<0, PL>([&](auto K_) {
                                constexpr int k = decltype(K_)::value;
                                W2[r][k] = op3<TT_MUX>(lt1, W1[r][k], key[k]);
                                W1[r][k] = op3<TT_MUX>(lt1, key[k], W1[r][k]);
                            });
                        } else {
                            const V lt2 = A::less_than(key, W2[r]);
                            sfor<0, PL>([&](auto K_) {
                                constexpr int k = decltype(K_)::value;
                                const V t = op3<TT_MUX>(lt2, key[k], W2[r][k]);
                                W2[r][k] = op3<TT_MUX>(lt1, W1[r][k], t);
                                W1[r][k] = op3<TT_MUX>(lt1, key[k], W1[r][k]);
                            });
                        }
                        sfor<0, ARG>([&](auto K_) {
                            constexpr int k = decltype(K_)::value;
                            argn[r][k] = ((slot >> k) & 1) ? B::or_(argn[r][k], lt1) : B::andn(argn[r][k], lt1);
                        });
                    }
                    // The instruction selector orders a basic block by its data flow alone and would compute values whose next use is
                    // an iteration away (the new v's zero-ness, the row minima) at the END of the block, holding their operands -- seven
                    // key planes per edge -- in registers until then.  An opaque use pins each update where it is written (rate 2/3).
                    pin_update(sv[e]); if constexpr (!GEO::nz_in_lds(e)) pin_update(nz[e]); pin_update(Sn[r]); pin_update(Pn[r]);
                    sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; pin_update(W1[r][k]); pin_update(W2[r][k]); });
                    sfor<0, ARG>([&](auto K_) { pin_update(argn[r][decltype(K_)::value]); });
                    if constexpr (e == GEO::last_edge(r) && !GEO::shared_row(r)) finish_row<r>();
                    B::fence();
                }
            });
        });
    }
    // ---- split mode: the rows shared with the other wave are finished after their exchange; returns the unsatisfied checks ----
    BS_FN V finish_iteration(B &)
    {
        sfor<0, NROWS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            if constexpr (GEO::shared_row(r)) finish_row<r>();
        });
        return fail;
    }
};

// ---- prologue / epilogue pieces shared by the one-wave and the split drivers ----------------------------------------------
// The 2048 raw LLR bytes of one block column of a group (M per codeword, G codewords) pass through a staging slab in LDS:
// position p = codeword * M + index.  The slab is SKEWED -- 16 bytes of padding after every 256 -- so that the byte gather below,
// whose lanes read at strides of Q and M, finds its 64 bytes in different banks (conflict-free for M >= 512, 2-way for TM1536, 4-way
// for TM1280, where every lane reads a dword of its own; unskewed: 4- to 8-way, all of SQ_LDS_BANK_CONFLICT of round 4's kernels).
constexpr int STAGE_BYTES = 2048 + 16 * 8;
constexpr int stage_skew(int p) { return p + 16 * (p / 256); }

template <int CODE, class B>
BS_FN void gather_planes(B &b, typename B::V p0, int stage, typename B::V (&X)[8]);

// block column c of the group's LLRs -> the lane's 8 bit planes X[p]: bit (index / L) of lane (q, index mod L) = bit p of the LLR.
// Global side: two 16-byte loads per lane, consecutive lanes consecutive bytes (the wave reads the column's 128-byte lines whole,
// once).  The loads are unconditional (a predicated load is an EXEC-masked branch per load): lanes of codewords beyond the batch read
// the group's FIRST frame, which exists, and the value is masked.
template <int CODE, class B, class D>
BS_FN void load_column_planes(B &b, const D &d, const int8_t *llrs, int c, uint32_t group, uint32_t batch, int stage, typename B::V (&X)[8])
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int M = GEO::M, N = GEO::N, Q = GEO::Q, W = GEO::W, G = GEO::G;
    sfor<0, 2>([&](auto H_) {
        constexpr int h = decltype(H_)::value;
        const V p = B::add(B::shl(d.lane, 4), B::c(1024 * h));                               // slab position of the lane's 16 bytes
        const V pcw = B::shr(p, ilog2c(M));                                                   // their codeword inside the group
        const V ok = B::less_u(B::add(B::c(group * (uint32_t)G), pcw), B::c(batch));
        const V src = B::add(B::mul_u(B::and_(pcw, ok), (uint32_t)N), B::add(B::c((uint32_t)c * M), B::and_(p, B::c(M - 1))));
        V w[4];
        b.gload128(llrs, src, w);
        sfor<0, 4>([&](auto I_) { constexpr int i = decltype(I_)::value; w[i] = B::and_(w[i], ok); });
        b.lds_write128(B::add(B::add(p, B::shl(B::shr(p, 8), 4)), B::c(stage)), w);          // stage_skew(p)
    });
    // lane (cw, q, ll) gathers the bytes of its 32 indices: q * Q + ll + L * bit; dword dd holds bits dd, 8 + dd, 16 + dd, 24 + dd
    const V cw = B::shr(d.lane, ilog2c(W));
    const V p0 = B::add(B::add(B::shl(cw, ilog2c(M)), B::shl(d.q, ilog2c(Q))), d.ll);       // position of bit 0
    // (compiler fences around the slab's use: the 16-byte vector stores above and the byte loads below are accesses of different types
    // to the same LDS bytes, and without the fence the NEXT use of the slab's stores were seen moved ahead of this use's loads -- round 6,
    // place_slot's two-piece staging: all-wrong decodes on the GPU, correct in the CPU emulation)
    B::mem_fence();
    gather_planes<CODE>(b, p0, stage, X);
    B::mem_fence();
}

// the 32 LLR bytes at slab positions p0 + L * bit (bit = 0 .. 31; the slab is skewed) -> the 8 bit planes X[p] of those 32 indices
template <int CODE, class B>
BS_FN void gather_planes(B &b, typename B::V p0, int stage, typename B::V (&X)[8])
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int L = GEO::L;
    auto at = [&](int bit) {                                                                  // skewed LDS address of bit `bit`
        if constexpr (L * 32 <= 256 && 256 % (L * 32) == 0) {
            // the lane's 32 bytes lie inside one 256-byte run: one skew for all of them
            return B::add(B::add(p0, B::shl(B::shr(p0, 8), 4)), B::c(stage + L * bit));
        } else {
            const V pp = B::add(p0, B::c(L * bit));
            return B::add(B::add(pp, B::shl(B::shr(pp, 8), 4)), B::c(stage));
        }
    };
    sfor<0, 8>([&](auto D_) {
        constexpr int dd = decltype(D_)::value;
        V x = b.lds_read_u8(at(dd));
        x = B::or_(x, B::shl(b.lds_read_u8(at(8 + dd)), 8));
        x = B::or_(x, B::shl(b.lds_read_u8(at(16 + dd)), 16));
        x = B::or_(x, B::shl(b.lds_read_u8(at(24 + dd)), 24));
        X[dd] = x;
    });
    // 8 x 8 bit-matrix transpose inside every byte lane: afterwards X[p] byte y bit dd = bit p of LLR (8 y + dd)
    auto stage_t = [&](auto S_, uint32_t mask) {
        constexpr int s = decltype(S_)::value;
        sfor<0, 8>([&](auto D_) {
            constexpr int dd = decltype(D_)::value;
            if constexpr ((dd & s) == 0) {
                const V t = B::and_(B::xor_(B::shr(X[dd], s), X[dd + s]), B::c(mask));
                X[dd + s] = B::xor_(X[dd + s], t);
                X[dd] = B::xor_(X[dd], B::shl(t, s));
            }
        });
    };
    stage_t(IC<4>{}, 0x0F0F0F0Fu);
    stage_t(IC<2>{}, 0x33333333u);
    stage_t(IC<1>{}, 0x55555555u);
}

// the hard-decision plane of one block column (64 words at LDS offset `base`, one per lane) -> the lane's dword of 32 consecutive
// output bits, MSB first inside each byte (decoder.rs:455-461)
template <int CODE, class B, class D>
BS_FN typename B::V pack_hard_column(B &b, const D &d, int base)
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int L = GEO::L;
    const V b0 = B::shl(d.ll, 5 - ilog2c(L));                                 // first bit of this lane's 32 indices: 32 ll / L
    const V qbase = B::shl(B::add(d.cwbase, B::shl(d.q, ilog2c(L))), 2);     // byte offset of the quarter's first lane
    V out = B::c(0);
    sfor<0, L>([&](auto LL_) {
        constexpr int l2 = decltype(LL_)::value;
        const V w = B::shr_v(b.lds_read32(B::add(qbase, B::c(base + 4 * l2))), b0);
        sfor<0, 32 / L>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            constexpr int t = l2 + L * k;                                     // index inside the 32, MSB first inside its byte
            constexpr int pos = 8 * (t / 8) + 7 - (t % 8);
            out = B::or_(out, B::shl(B::and_(B::shr(w, k), B::c(1)), pos));
        });
    });
    return out;
}

// once per kernel (per wave): the lane permutation tables
template <int CODE, class B>
BS_FN void init_kernel(B &b)
{
    Decoder<CODE, B> d;
    d.init_lane(b);
    d.init_perm_tables(b);
}

// ---- driver: one wave decodes group after group of G codewords ---------------------------------------------------------------
// llrs [batch][N] i8, output [batch][NP/8] MSB first, iters [batch], success [batch]; `group` = index of the group of G frames.
// (Lane offsets are relative to the group's first frame, so they fit 32 bits whatever the batch.)
template <int CODE, class B>
BS_FN void decode_group(B &b, const int8_t *llrs_all, uint8_t *output_all, uint32_t *iters_all, uint8_t *success_all, uint32_t batch,
                        uint32_t maxiters, uint32_t group)
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int M = GEO::M, W = GEO::W, G = GEO::G, NTX = GEO::NTX, NCOLS = GEO::NCOLS;
    Decoder<CODE, B> d;
    d.init_lane(b);
    const V lane = d.lane;
    const V cw = B::shr(lane, ilog2c(W));                                    // codeword of the lane inside the group (0 for W = 64)
    const int8_t *llrs = llrs_all + (size_t)group * G * GEO::N;
    uint8_t *output = output_all + (size_t)group * G * GEO::OUT_LEN;
    uint32_t *iters = iters_all + (size_t)group * G;
    uint8_t *success = success_all + (size_t)group * G;
    const V valid = B::less_u(B::add(B::c(group * (uint32_t)G), cw), B::c(batch));      // all ones where the lane's codeword exists
    const uint64_t valid_mask = b.ballot(valid);

    // ---- LLRs: 2048 raw bytes of block column c (32 per lane) -> staging slab -> 8 bit planes per lane ----
    sfor<0, NTX>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        V X[8];
        load_column_planes<CODE>(b, d, llrs, c, group, batch, GEO::LDS_STAGE, X);
        sfor<0, 8>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            b.lds_write32(B::add(B::shl(lane, 2), B::c(GEO::LDS_LLR + (c * LLRP + k) * 256)), X[k]);
        });
    });

    d.reset_state(b);                  // (after the LLRs: the hard-decision words alias the staging slab)
    d.prime_perm(b);

    // ---- iterations (decoder.rs:380-464): the codewords of the wave run in lockstep, a finished one is frozen.  Verdicts are
    // wave-uniform per codeword: iteration counts and success flags live in scalars, not in planes ----
    uint64_t frozen_mask = ~valid_mask, ok_mask = 0;
    uint32_t iters_s[G];
    sfor<0, G>([&](auto G_) { iters_s[decltype(G_)::value] = maxiters; });
    for (uint32_t it = 0; it < maxiters && frozen_mask != ~0ull; ++it) {
        const V fail = d.iteration(b, frozen_mask);
        const uint64_t unsat_lanes = b.ballot(fail);
        sfor<0, G>([&](auto G_) {
            constexpr int g = decltype(G_)::value;
            constexpr uint64_t gm = (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1)) << ((g * W) & 63);
            if (!(unsat_lanes & gm) && !(frozen_mask & gm)) {                  // satisfied for the first time: (true, it)  (:453-463)
                iters_s[g] = it;
                ok_mask |= gm;
                frozen_mask |= gm;
            }
        });
    }
    V iters_v = B::c(iters_s[0]);
    sfor<1, G>([&](auto G_) {
        constexpr int g = decltype(G_)::value;
        constexpr uint64_t gm = ((1ull << (W & 63)) - 1) << ((g * W) & 63);
        iters_v = B::select_lanes(gm, B::c(iters_s[g]), iters_v);
    });
    const V ok_v = B::and_(b.plane_of(ok_mask), B::c(1));

    // ---- hard decisions, MSB first (decoder.rs:455-461 / :467-473): plane -> LDS -> one dword of 32 consecutive bits per lane ----
    if constexpr (!GEO::HARD_LDS)
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            b.lds_write32(d.hard_addr(c), d.hard[c]);
        });
    // (the lane constants are formed again from an opaque copy of the lane index: as common subexpressions of the prologue's they would
    // hold half a dozen registers through the iteration loop)
    d.reinit_lane();
    const V cw2 = B::shr(d.lane, ilog2c(W)), lw2 = B::and_(d.lane, B::c(W - 1));
    const V valid2 = B::less_u(B::add(B::c(group * (uint32_t)G), cw2), B::c(batch));
    sfor<0, NCOLS>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        const V out = pack_hard_column<CODE>(b, d, GEO::LDS_HARD + GEO::col_slot(c) * 256);
        // (the kernels that keep spilled values in scratch stream their output past the L2 as well: default-policy stores allocate
        // lines there and cost TM8192 4 % and TM2048 3 %; profiles/r05_kbench/spill_leak.txt)
        if constexpr (GEO::HARD_LDS) b.gstore32_stream(output, B::add(B::mul_u(cw2, (uint32_t)GEO::OUT_LEN), B::add(B::c((uint32_t)c * (M / 8)), B::shl(lw2, 2))), out, valid2);
        else b.gstore32(output, B::add(B::mul_u(cw2, (uint32_t)GEO::OUT_LEN), B::add(B::c((uint32_t)c * (M / 8)), B::shl(lw2, 2))), out, valid2);
    });
    const V first = B::and_(valid2, B::eq(lw2, B::c(0)));
    b.gstore32(iters, B::shl(cw2, 2), iters_v, first);
    b.gstore8(success, cw2, ok_v, first);
}


// ---- slot refill (round 6) -------------------------------------------------------------------------------------------------------
// The reference's harness hands a worker its next frame the moment one finishes (perftest/src/main.rs:39-52).  decode_group holds all
// G codewords of a wave until the slowest is done: 1.05-1.19 x the frames' own passes for G <= 4, 1.31 x for TM1536 (G = 8).  Here a
// slot whose codeword is finished -- converged, or at the iteration cap -- hands in its results and takes the wave's next frame while
// the other slots keep iterating.  Round 5 built this with whole-wave prologues / epilogues (every block column of every slot, masked)
// and lost: an event cost half an iteration (profiles/r05_kbench/slot_refill.txt).  Here an event is PER SLOT and COOPERATIVE: all 64
// lanes work on the one codeword that leaves (its NCOLS x W output dwords, one per lane) and on the one that arrives (its N LLR bytes
// through the staging slab, then one (block column, lane) unit of 32 LLRs per lane: the byte gather and the 8 x 8 bit transposes run
// once per slot instead of once per block column), and only the state RESET touches every state register (one v_cndmask each).
// Frames are wave-uniform scalars, so a slot's global addresses are a scalar base plus a lane offset whatever the batch size.
template <int G, int W>
struct Slots {
    static constexpr uint64_t gm(int g) { return (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1)) << ((g * W) & 63); }      // the lanes of slot g
    uint64_t active = 0;                   // lanes of the slots with a codeword in progress
    uint64_t fin = ~0ull, ok = 0;          // lanes of the slots that are finished (initially: all are to be filled); of those, the successes
    uint32_t it[G], fr[G];                 // iterations done, frame per slot
    BS_FN void start() { active = 0; fin = ~0ull; ok = 0; sfor<0, G>([&](auto G_) { it[decltype(G_)::value] = 0; fr[decltype(G_)::value] = 0; }); }
    // slots at the iteration cap are finished as they are: (false, max_iters)  (decoder.rs:466-474; max_iters = 0: at once).
    // Branch-free, like verdict(): these run inside the iteration loop, where scalar branches cut the loop body into blocks across
    // which the compiler keeps 70 more values live (round 5)
    BS_FN void expire(uint32_t maxiters)
    {
        sfor<0, G>([&](auto G_) { constexpr int g = decltype(G_)::value; fin |= ((active & gm(g)) != 0 && it[g] >= maxiters) ? gm(g) : 0ull; });
    }
    // after an iteration: a slot whose checks all hold is finished with (true, its iteration index)  (:453-463); the others count on
    BS_FN void verdict(uint64_t unsat_lanes)
    {
        sfor<0, G>([&](auto G_) {
            constexpr int g = decltype(G_)::value;
            const bool act = (active & gm(g)) != 0, sat = (unsat_lanes & gm(g)) == 0;
            const uint64_t won = (act && sat) ? gm(g) : 0ull;
            fin |= won;
            ok |= won;
            it[g] += (act && !sat) ? 1u : 0u;
        });
    }
};

// the finished codeword of the slot whose first lane is `first`: its hard-decision words (LDS, one per block column and lane) -> the
// NCOLS x W dwords of its output, MSB first inside each byte (decoder.rs:455-461 / :467-473), one dword per lane and round
template <int CODE, class B, class D>
BS_FN void emit_slot(B &b, const D &d, int first, uint8_t *out, uint32_t *iters_p, uint8_t *ok_p, uint32_t iters_v, uint32_t ok_v)
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int M = GEO::M, L = GEO::L, W = GEO::W, NCOLS = GEO::NCOLS, UNITS = NCOLS * W;
    sfor<0, (UNITS + 63) / 64>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        const V u = B::add(d.lane, B::c(64 * r));
        const V valid = B::less_u(u, B::c(UNITS));
        const V c = B::and_(B::shr(u, ilog2c(W)), valid);                      // block column (0 for the idle lanes of the last round)
        const V lw = B::and_(u, B::c(W - 1));                                     // lane of the codeword
        const V b0 = B::shl(B::and_(lw, B::c(L - 1)), 5 - ilog2c(L));            // first bit of this unit's 32 indices: 32 ll / L
        const V qbase = B::add(B::shl(c, 8), B::shl(B::add(B::c(first), B::and_(lw, B::c(~(L - 1) & (W - 1)))), 2));   // column's words, quarter's first lane
        V o = B::c(0);
        sfor<0, L>([&](auto LL_) {
            constexpr int l2 = decltype(LL_)::value;
            const V w = B::shr_v(b.lds_read32(B::add(qbase, B::c(GEO::LDS_HARD + 4 * l2))), b0);
            sfor<0, 32 / L>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                constexpr int t = l2 + L * k;
                constexpr int pos = 8 * (t / 8) + 7 - (t % 8);
                o = B::or_(o, B::shl(B::and_(B::shr(w, k), B::c(1)), pos));
            });
        });
        const V off = B::add(B::mul_u(c, (uint32_t)(M / 8)), B::shl(lw, 2));
        if constexpr (GEO::HARD_LDS) b.gstore32_stream(out, off, o, valid);
        else b.gstore32(out, off, o, valid);
    });
    const V first_lane = B::eq(d.lane, B::c(0));
    b.gstore32(iters_p, B::c(0), B::c(iters_v), first_lane);
    b.gstore8(ok_p, B::c(0), B::c(ok_v), first_lane);
}

// a fresh codeword for the slot whose first lane is `first`: its N LLR bytes -> staging slab (16 bytes per lane and round, whole
// lines) -> per lane ONE unit (block column c, codeword lane lw) of 32 LLRs -> its 8 bit planes, written to that lane's plane words.
// The slab is the hard-decision words' area; where the "v != 0" planes of the ACTIVE slots live behind those words (Geo::NZ_IN_SLAB:
// rate 1/2) only the words' own NCOLS x 256 bytes are free, and the codeword passes through in pieces of 1024 bytes.
template <int CODE> struct SlotLoad {
    using GEO = Geo<CODE>;
    static constexpr int M = GEO::M, N = GEO::N, W = GEO::W;
    static constexpr int SLAB_FREE = GEO::NZ_IN_SLAB > 0 ? GEO::NCOLS * 256 : GEO::HARD_BYTES;
#ifdef BS_REFILL_PIECE
    static constexpr int PIECE = (N % BS_REFILL_PIECE == 0 && BS_REFILL_PIECE % M == 0) ? BS_REFILL_PIECE : N;
#else
    static constexpr int PIECE = (N + 16 * (N / 256) <= SLAB_FREE) ? N : 1024;      // bytes staged at a time
#endif
    static_assert(PIECE % M == 0 && N % PIECE == 0 && PIECE % 16 == 0 && PIECE + 16 * (PIECE / 256) <= SLAB_FREE, "a piece is whole block columns and fits the slab");
    static constexpr int COLS = PIECE / M, UNITS = COLS * W;                         // block columns and (column, lane) units per piece
    static constexpr int ROUNDS = (PIECE + 1023) / 1024, PIECES = N / PIECE;        // 16-byte loads per lane and piece; pieces
    // position inside the piece of the lane's 16 bytes of round r (the last round's idle lanes repeat the piece's last 16 bytes)
    template <int R, class B> static BS_FN typename B::V pos(typename B::V lane)
    {
        using V = typename B::V;
        V p = B::add(B::shl(lane, 4), B::c(1024 * R));
        if constexpr (1024 * (R + 1) > PIECE) {
            const V over = B::less_u(B::c(PIECE - 16), p);
            p = B::template bitop3<TT_MUX>(over, B::c(PIECE - 16), p);
        }
        return p;
    }
};

// the N LLR bytes of a frame into registers: 16 bytes per lane, round and piece (whole lines, once).  Issued one frame AHEAD of
// their use (decode_refill): an event then finds its codeword's bytes arrived long ago instead of waiting an HBM round trip.
template <int CODE, class B, class D>
BS_FN void fetch_slot(B &b, const D &d, const int8_t *src, typename B::V (&w)[SlotLoad<CODE>::PIECES][SlotLoad<CODE>::ROUNDS][4])
{
    using SL = SlotLoad<CODE>;
    sfor<0, SL::PIECES>([&](auto P_) {
        constexpr int piece = decltype(P_)::value;
        sfor<0, SL::ROUNDS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            b.gload128(src, B::add(SL::template pos<r, B>(d.lane), B::c(piece * SL::PIECE)), w[piece][r]);
        });
    });
}

template <int CODE, class B, class D>
BS_FN void place_slot(B &b, const D &d, int first, const typename B::V (&w)[SlotLoad<CODE>::PIECES][SlotLoad<CODE>::ROUNDS][4])
{
    using GEO = Geo<CODE>;
    using SL = SlotLoad<CODE>;
    using V = typename B::V;
    constexpr int M = GEO::M, L = GEO::L, Q = GEO::Q, W = GEO::W, COLS = SL::COLS, UNITS = SL::UNITS;
    sfor<0, SL::PIECES>([&](auto P_) {
        constexpr int piece = decltype(P_)::value;
        sfor<0, SL::ROUNDS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V p = SL::template pos<r, B>(d.lane);
            b.lds_write128(B::add(B::add(p, B::shl(B::shr(p, 8), 4)), B::c(GEO::LDS_STAGE)), w[piece][r]);       // stage_skew(p)
        });
        B::mem_fence();            // (the slab's vector stores and byte loads: see load_column_planes)
        sfor<0, (UNITS + 63) / 64>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V u = B::add(d.lane, B::c(64 * r));
            const V valid = B::less_u(u, B::c(UNITS));
            const V cl = B::and_(B::shr(u, ilog2c(W)), valid);                        // block column inside the piece
            const V lw = B::and_(u, B::c(W - 1));
            // slab position of bit 0 of unit (cl, q, ll): cl * M + q * Q + ll
            const V p0 = B::add(B::add(B::shl(cl, ilog2c(M)), B::shl(B::shr(lw, ilog2c(L)), ilog2c(Q))), B::and_(lw, B::c(L - 1)));
            V X[8];
            gather_planes<CODE>(b, p0, GEO::LDS_STAGE, X);
            const V at = B::add(B::mul_u(B::add(cl, B::c(piece * COLS)), (uint32_t)(LLRP * 256)), B::shl(B::add(B::c(first), lw), 2));
            sfor<0, 8>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                b.lds_write32_if(B::add(at, B::c(GEO::LDS_LLR + k * 256)), X[k], valid);
            });
        });
        B::mem_fence();
    });
}

// ---- driver: one wave decodes frame after frame, G at a time, a finished slot taking the next frame `next()` hands out (a frame index,
// or NO_FRAME once the supply is exhausted: the slots then drain).  llrs [batch][N] i8, output [batch][NP/8] MSB first, iters, success.
constexpr uint32_t NO_FRAME = 0xFFFFFFFFu;
template <int CODE, class B, class NEXT>
BS_FN void decode_refill(B &b, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, uint32_t maxiters, NEXT next)
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    constexpr int W = GEO::W, G = GEO::G, NCOLS = GEO::NCOLS;
    static_assert(!GEO::SPLIT && G >= 2, "slot refill: the one-wave kernels with several codewords per wave");
    Decoder<CODE, B> d;
    d.init_lane(b);
    d.prime_perm(b);
    Slots<G, W> s;
    s.start();
    d.reset_state(b);
    // one frame of look-ahead: its LLR bytes are requested when its predecessor is placed and sit in registers until a slot is free
    V ahead[SlotLoad<CODE>::PIECES][SlotLoad<CODE>::ROUNDS][4];
    uint32_t ahead_frame = next();
    if (ahead_frame != NO_FRAME) fetch_slot<CODE>(b, d, llrs + (size_t)ahead_frame * GEO::N, ahead);
    // Two loops: the outer one runs once per EVENT -- some slots finished --, the inner one is the iteration loop proper, with nothing
    // of the refill inside it (one loop with the refill as a branch cost the register allocation of the hot path dearly: round 5).
    for (;;) {
        d.reinit_lane();       // (lane constants formed here, from an opaque copy of the lane index: as common subexpressions of an earlier
                               // block's they would hold registers through the iterations)
        const uint64_t done = s.fin & s.active;
        if constexpr (!GEO::HARD_LDS)
            if (done) sfor<0, NCOLS>([&](auto C_) { constexpr int c = decltype(C_)::value; b.lds_write32(d.hard_addr(c), d.hard[c]); });
        uint64_t fresh = 0;
        sfor<0, G>([&](auto G_) {
            constexpr int g = decltype(G_)::value;
            constexpr uint64_t gm = Slots<G, W>::gm(g);
            if (s.fin & gm) {
                if (s.active & gm) {
                    const size_t f = s.fr[g];
                    emit_slot<CODE>(b, d, g * W, output + f * GEO::OUT_LEN, iters + f, success + f, s.it[g], (s.ok & gm) ? 1u : 0u);
                }
                s.active &= ~gm;
            }
        });
        // (the arrivals after ALL departures: the staging slab is the hard-decision words' area)
        sfor<0, G>([&](auto G_) {
            constexpr int g = decltype(G_)::value;
            constexpr uint64_t gm = Slots<G, W>::gm(g);
            if ((s.fin & gm) && ahead_frame != NO_FRAME) {
                place_slot<CODE>(b, d, g * W, ahead);
                s.fr[g] = ahead_frame; s.it[g] = 0; s.active |= gm; fresh |= gm;
                ahead_frame = next();
                if (ahead_frame != NO_FRAME) fetch_slot<CODE>(b, d, llrs + (size_t)ahead_frame * GEO::N, ahead);
            }
        });
        if (fresh) d.reset_state(b, fresh);    // (after the LLRs: the hard-decision words alias the staging slab)
        s.fin = 0; s.ok = 0;
        s.expire(maxiters);                    // (max_iters = 0: the fresh slots are at the cap already)
        if (s.fin) continue;
        if (!s.active) break;
        // ---- iterations of every slot (decoder.rs:380-464) until one of them finishes; lanes of empty slots compute on whatever they hold ----
        do {
            const V fail = d.iteration(b, ~s.active);
            s.verdict(b.ballot(fail));
            s.expire(maxiters);
        } while (!s.fin);
    }
}

}  // namespace bs
}  // namespace ldpc
