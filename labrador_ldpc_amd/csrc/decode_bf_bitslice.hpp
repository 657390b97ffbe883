// decode_bf_bitslice.hpp -- the hard-decision decoder (LDPCCode::decode_bf, /root/reference/src/decoder.rs:243-301, with its erasure
// pre-pass decode_erasures, :144-223) in the bit-sliced layout of decode_ms_bitslice.hpp, for the TM codes.
//
// A hard decision IS one bit: a block column of a codeword is one 32-bit register per lane (M/32 lanes per codeword), a check row's
// parity is one register, the per-variable violation count (at most 6: the largest variable degree) three.  One iteration of
// decoder.rs:264-298 is then, per edge, one XOR into the row's parity and one 3-bit ripple increment of the variable's count
// (plus ds_bpermute + v_alignbit for the pi_k blocks, each way) -- ~12 instructions per 2048 edges -- and the "flip every bit that
// has the maximum count" step is a ballot per count value.  The byte-per-variable kernel of decode_bf.hip walks the same edges
// with one LDS byte access and one LDS atomic per edge.
//
// Reference behaviour reproduced deliberately (as decode_bf.hip does, see there): the erasure pass always stops after its FIRST
// iteration with (true, 0) because it counts every still-erased variable into `bits_fixed` (decoder.rs:205-213); a check's erasure
// count is its number of edges into the punctured column (all punctured bits start erased); a punctured bit becomes 1 iff the
// checks that have it as their ONLY erased variable vote so by majority.  Returned iterations = bit-flipping iterations (+ 0).
// Same backend abstraction as decode_ms_bitslice.hpp: tests/c/bitslice_emu.cpp runs this text on the CPU against the oracle.
#pragma once

#include "decode_ms_bitslice.hpp"

namespace ldpc {
namespace bs {

template <int CODE>
struct BfGeo {
    using GEO = Geo<CODE>;
    static constexpr int NX = GEO::NX, NCOLS = GEO::NCOLS, L = GEO::L;
    // LDS: lane permutations [NX][2][64] words, one exchange area of max(L, NCOLS) x 64 words (bit scatter on load, planes on store)
    static constexpr int LDS_PERM = 0, LDS_X = NX * 2 * 256, X_WORDS = (L > NCOLS ? L : NCOLS) * 64, LDS_BYTES = LDS_X + X_WORDS * 4;
    static constexpr int punctured_edges_in_row(int row)
    {
        int c = 0;
        for (int b = 0; b < GEO::NB; ++b) c += (GEO::P.blk[b].row == row && GEO::P.blk[b].col >= GEO::NTX) ? 1 : 0;
        return c;
    }
};

template <int CODE, class B>
BS_FN void bf_init_kernel(B &b)
{
    using GEO = Geo<CODE>;
    using V = typename B::V;
    Decoder<CODE, B> d;
    d.init_lane(b);
    sfor<0, GEO::NB>([&](auto E_) {
        constexpr int e = decltype(E_)::value;
        if constexpr (!GEO::local(e)) {
            V addr, amt;
            d.template perm_c2v<GEO::P.blk[e].val>(addr, amt);
            b.lds_write32(B::add(B::shl(d.lane, 2), B::c(BfGeo<CODE>::LDS_PERM + (GEO::exch_of(e) * 2 + 0) * 256)), B::or_(addr, B::shl(B::and_(amt, B::c(31)), 8)));
            d.template perm_v2c<GEO::P.blk[e].val>(addr, amt);
            b.lds_write32(B::add(B::shl(d.lane, 2), B::c(BfGeo<CODE>::LDS_PERM + (GEO::exch_of(e) * 2 + 1) * 256)), B::or_(addr, B::shl(B::and_(amt, B::c(31)), 8)));
        }
    });
}

// input [batch][N/8] hard bits MSB first, output [batch][NP/8], iters [batch], success [batch]
template <int CODE, class B>
BS_FN void bf_decode_group(B &b, const uint8_t *input_all, uint8_t *output_all, uint32_t *iters_all, uint8_t *success_all, uint32_t batch,
                           uint32_t maxiters, uint32_t group)
{
    using GEO = Geo<CODE>;
    using BG = BfGeo<CODE>;
    using V = typename B::V;
    constexpr int M = GEO::M, N = GEO::N, L = GEO::L, W = GEO::W, G = GEO::G, NTX = GEO::NTX, NCOLS = GEO::NCOLS, NB = GEO::NB, NROWS = GEO::NROWS;
    constexpr int PER = 32 / L;                                                // bits a lane contributes to (takes from) one 32-index dword
    auto op3 = [](auto TT_, V x, V y, V z) { return B::template bitop3<decltype(TT_)::value>(x, y, z); };
    Decoder<CODE, B> d;
    d.init_lane(b);
    const V lane = d.lane;
    const V cw = B::shr(lane, ilog2c(W)), lw = B::and_(lane, B::c(W - 1));
    const uint8_t *input = input_all + (size_t)group * G * (N / 8);
    uint8_t *output = output_all + (size_t)group * G * GEO::OUT_LEN;
    uint32_t *iters = iters_all + (size_t)group * G;
    uint8_t *success = success_all + (size_t)group * G;
    const V valid = B::less_u(B::add(B::c(group * (uint32_t)G), cw), B::c(batch));
    const uint64_t valid_mask = b.ballot(valid);
    const V b0 = B::shl(d.ll, 5 - ilog2c(L));                                  // first bit of the lane's 32 consecutive indices: 32 ll / L
    const V qbase = B::shl(B::add(d.cwbase, B::shl(d.q, ilog2c(L))), 2);      // byte offset of the quarter's first lane in a 64-word row
    auto perm = [&](int idx) { return b.lds_read32(B::add(B::shl(lane, 2), B::c(BG::LDS_PERM + idx * 256))); };
    auto pull = [&](V entry, V x) { return B::rotr(b.bperm(entry, x), B::shr(entry, 8)); };

    // ---- hard bits -> one plane per block column: the lane's dword of 32 consecutive bits is dealt to the L lanes of its quarter ----
    V bits[NCOLS];
    sfor<0, NCOLS>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        if constexpr (c >= NTX) { bits[c] = B::c(0); return; }                 // punctured bits start at 0 (decoder.rs:167)
        else {
            // bytes MSB first: index t of the 32 sits at bit 8 (t / 8) + 7 - t % 8 of the little-endian dword
            const V w = B::and_(b.gload32(input, B::add(B::mul_u(cw, (uint32_t)(N / 8)), B::add(B::c((uint32_t)c * (M / 8)), B::shl(lw, 2))), valid), valid);
            sfor<0, L>([&](auto LL_) {
                constexpr int l2 = decltype(LL_)::value;
                V piece = B::c(0);                                             // bits k = 0 .. PER - 1: index t = l2 + L k
                sfor<0, PER>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    constexpr int t = l2 + L * k, pos = 8 * (t / 8) + 7 - (t % 8);
                    piece = B::or_(piece, B::shl(B::and_(B::shr(w, pos), B::c(1)), k));
                });
                // to lane (q, l2) of this quarter, slot = this lane's ll: row ll of the exchange area, column of the target lane
                b.lds_write32(B::add(B::add(B::shl(d.ll, 8), qbase), B::c(BG::LDS_X + 4 * l2)), B::shl_v(piece, b0));
            });
            V acc = B::c(0);
            sfor<0, L>([&](auto S_) { acc = B::or_(acc, b.lds_read32(B::add(B::shl(lane, 2), B::c(BG::LDS_X + decltype(S_)::value * 256)))); });
            bits[c] = acc;
        }
    });

    // ---- erasure pre-pass (decoder.rs:144-223): one effective iteration ----
    if (maxiters > 0) {
        sfor<NTX, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            V pos[3] = {B::c(0), B::c(0), B::c(0)}, neg[3] = {B::c(0), B::c(0), B::c(0)};
            sfor<0, NB>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                constexpr int r = GEO::P.blk[e].row;
                if constexpr (GEO::P.blk[e].col == c && BG::punctured_edges_in_row(r) == 1) {           // exactly one erasure (:194)
                    V par = B::c(0);                                                                 // over the non-erased bits (:185-188)
                    sfor<0, NB>([&](auto F_) {
                        constexpr int f = decltype(F_)::value;
                        if constexpr (GEO::P.blk[f].row == r && GEO::P.blk[f].col < NTX) {
                            if constexpr (GEO::local(f)) par = B::xor_(par, bits[GEO::P.blk[f].col]);
                            else par = B::xor_(par, pull(perm(GEO::exch_of(f) * 2 + 1), bits[GEO::P.blk[f].col]));
                        }
                    });
                    V pv = par;                                                                       // at the variable's alignment
                    if constexpr (!GEO::local(e)) pv = pull(perm(GEO::exch_of(e) * 2 + 0), par);
                    auto inc = [&](V (&cnt)[3], V x) {                                               // 3-bit ripple increment where x
                        const V c1 = B::and_(cnt[0], x); cnt[0] = B::xor_(cnt[0], x);
                        const V c2 = B::and_(cnt[1], c1); cnt[1] = B::xor_(cnt[1], c1);
                        cnt[2] = B::xor_(cnt[2], c2);
                    };
                    inc(pos, pv);                                                                     // +1 (:196-197)
                    inc(neg, B::not_(pv));                                                            // -1 (:198-199)
                }
            });
            // votes > 0  <=>  neg < pos (3-bit borrow chain)
            V br = B::andn(pos[0], neg[0]);
            br = op3(IC<TT_BORROW>{}, pos[1], neg[1], br);
            br = op3(IC<TT_BORROW>{}, pos[2], neg[2], br);
            bits[c] = br;                                                                             // :207-210
        });
    }

    // ---- bit flipping (decoder.rs:264-298) ----
    uint64_t done_mask = ~valid_mask;
    V iters_v = B::c(maxiters), ok_v = B::c(0);
    for (uint32_t it = 0; it < maxiters && done_mask != ~0ull; ++it) {
        V par[NROWS];
        sfor<0, NROWS>([&](auto R_) { par[decltype(R_)::value] = B::c(0); });
        sfor<0, NB>([&](auto E_) {                                                                    // :269-273
            constexpr int e = decltype(E_)::value;
            constexpr int r = GEO::P.blk[e].row, c = GEO::P.blk[e].col;
            if constexpr (GEO::local(e)) par[r] = B::xor_(par[r], bits[c]);
            else par[r] = B::xor_(par[r], pull(perm(GEO::exch_of(e) * 2 + 1), bits[c]));
        });
        V cnt[NCOLS][3];
        sfor<0, NCOLS>([&](auto C_) {                                                                 // :277-286
            constexpr int c = decltype(C_)::value;
            cnt[c][0] = B::c(0); cnt[c][1] = B::c(0); cnt[c][2] = B::c(0);
            sfor<0, NB>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                if constexpr (GEO::P.blk[e].col == c) {
                    constexpr int r = GEO::P.blk[e].row;
                    V x = par[r];
                    if constexpr (!GEO::local(e)) x = pull(perm(GEO::exch_of(e) * 2 + 0), par[r]);
                    const V c1 = B::and_(cnt[c][0], x); cnt[c][0] = B::xor_(cnt[c][0], x);
                    const V c2 = B::and_(cnt[c][1], c1); cnt[c][1] = B::xor_(cnt[c][1], c1);
                    cnt[c][2] = B::xor_(cnt[c][2], c2);
                }
            });
        });
        // max_violations per codeword (:276, :282-284): for v = 7 .. 1, which lanes hold a variable with exactly v violations?
        uint64_t mx_bit[3] = {0, 0, 0}, found = 0;                       // per lane: the bits of its codeword's maximum; found = maximum known
        for (int v = 7; v >= 1; --v) {
            V any = B::c(0);
            sfor<0, NCOLS>([&](auto C_) {
                constexpr int c = decltype(C_)::value;
                const V e0 = (v & 1) ? cnt[c][0] : B::not_(cnt[c][0]), e1 = (v & 2) ? cnt[c][1] : B::not_(cnt[c][1]), e2 = (v & 4) ? cnt[c][2] : B::not_(cnt[c][2]);
                any = B::or_(any, op3(IC<TT_AND3>{}, e0, e1, e2));
            });
            const uint64_t lanes = b.ballot(any);
            uint64_t has = 0;
            if constexpr (W == 64) has = lanes ? ~0ull : 0ull;
            else {
                constexpr uint64_t gm = (1ull << W) - 1;
                for (int g = 0; g < G; ++g)
                    if ((lanes >> (g * W)) & gm) has |= gm << (g * W);
            }
            const uint64_t fresh = has & ~found;
            found |= fresh;
            for (int k = 0; k < 3; ++k)
                if ((v >> k) & 1) mx_bit[k] |= fresh;
        }
        const uint64_t newly = ~done_mask & ~found;                        // max_violations == 0: (true, it)  (:288-289)
        const V nw = b.plane_of(newly);
        iters_v = B::template bitop3<TT_MUX>(nw, B::c(it), iters_v);
        ok_v = B::or_(ok_v, B::and_(nw, B::c(1)));
        done_mask |= newly;
        // flip every bit whose count equals its codeword's maximum (:292-296); finished codewords are left alone
        const V live = b.plane_of(~done_mask);
        const V m0 = b.plane_of(mx_bit[0]), m1 = b.plane_of(mx_bit[1]), m2 = b.plane_of(mx_bit[2]);
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            const V ne = op3(IC<TT_OR3>{}, B::xor_(cnt[c][0], m0), B::xor_(cnt[c][1], m1), B::xor_(cnt[c][2], m2));      // count != maximum
            bits[c] = B::xor_(bits[c], B::andn(live, ne));
        });
    }

    // ---- planes -> bytes, MSB first ----
    sfor<0, NCOLS>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        b.lds_write32(B::add(B::shl(lane, 2), B::c(BG::LDS_X + c * 256)), bits[c]);
    });
    sfor<0, NCOLS>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        V out = B::c(0);
        sfor<0, L>([&](auto LL_) {
            constexpr int l2 = decltype(LL_)::value;
            const V w = B::shr_v(b.lds_read32(B::add(qbase, B::c(BG::LDS_X + c * 256 + 4 * l2))), b0);
            sfor<0, PER>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                constexpr int t = l2 + L * k, pos = 8 * (t / 8) + 7 - (t % 8);
                out = B::or_(out, B::shl(B::and_(B::shr(w, k), B::c(1)), pos));
            });
        });
        b.gstore32(output, B::add(B::mul_u(cw, (uint32_t)GEO::OUT_LEN), B::add(B::c((uint32_t)c * (M / 8)), B::shl(lw, 2))), out, valid);
    });
    const V first = B::and_(valid, B::eq(lw, B::c(0)));
    b.gstore32(iters, B::shl(cw, 2), iters_v, first);
    b.gstore8(success, cw, ok_v, first);
}

}  // namespace bs
}  // namespace ldpc
