// decode_ms_bitslice_split.hpp -- the bit-sliced i8 min-sum decoder of decode_ms_bitslice.hpp with a group of codewords shared by TWO
// waves (decode_ms::<i8>, /root/reference/src/decoder.rs:42-50, :347-475), for the rate-4/5 codes (TM1280, TM5120).
//
// Why.  A rate-4/5 codeword group has 39 edges: 218 planes of state before any temporary.  One wave holds that in 256 registers only
// on a diet (decode_ms_bitslice.hpp: LLR planes in a global workspace re-read in every iteration, block row 0 and the hard decisions
// in LDS, every use pinned in place): 164 VALU instructions per edge against 135 on the rate-1/2 codes, 12-22 x the algorithmic bytes at
// the L2 boundary.  Here wave 0 owns a set of block columns (Geo::SPLIT_MASK) with all their edges and wave 1 the rest (19 + 20 edges, five
// transmitted columns each): a wave's state is ~120 planes, its LLR planes fit LDS (10 KB), nothing is pinned or spilled, and the
// launch moves its algorithmic bytes.
//
// What the two waves exchange.  An edge's message u needs its block row's state -- the two smallest |v| of the row, the sign product,
// the arg-min slot -- over ALL the row's edges, and a row's edges lie in both halves (block rows 1 and 2; row 0's three edges are all
// wave 1's).  Each wave accumulates the row state over ITS edges.  Once per iteration wave 0 publishes its partial state of row 1 and
// wave 1 of row 2 (a 24-plane LDS buffer per wave); after a barrier each wave merges the row the OTHER published into its own partial
// state of that row (the two smallest keys of a union are min(a1, b1) and min(max(a1, b1), the winner's min2); signs and parities XOR; the
// arg-min slot follows the smaller min1, wave 0's on a tie), finishes the row (minima back to magnitudes: the row's new OLD state) and
// writes that into the other's buffer; after a second barrier each wave fetches the finished state of the row it published.  Both waves then hold the same state.  One merge per wave and
// two workgroup barriers per iteration (two waves with balanced work: they arrive together).
//
// The stages are separate member functions so that tests/c/bitslice_emu.cpp can run the two halves of a group alternately on the CPU.
#pragma once

#include "decode_ms_bitslice.hpp"

namespace ldpc {
namespace bs {

// LDS of a split workgroup: [wave 0: permutations | LLR planes] [wave 1: the same] [exchange buffer of wave 0] [of wave 1]
template <int CODE>
struct SplitLayout {
    using G0 = Geo<CODE, 0>;
    using G1 = Geo<CODE, 1>;
    static constexpr int XEXTRA = 2 * PL + 2 + G0::ARG;                       // W1 PL, W2 PL, Sn, Pn, argn ARG; then the parity of the unshared rows
    static constexpr int XPLANES = XEXTRA + 1;
    static constexpr int XBYTES = XPLANES * 256;
    static_assert(XBYTES >= STAGE_BYTES, "the exchange buffer doubles as the staging slab of the LLR transposition");
    static constexpr int PRIV0 = 0, PRIV1 = G0::LDS_PRIVATE, XBUF0 = PRIV1 + G1::LDS_PRIVATE, XBUF1 = XBUF0 + XBYTES, BYTES = XBUF1 + XBYTES;
    template <int H> static constexpr int priv() { return H == 0 ? PRIV0 : PRIV1; }
    template <int H> static constexpr int own_x() { return (H == 0 ? XBUF0 : XBUF1) - priv<H>(); }      // relative to the wave's private base
    template <int H> static constexpr int other_x() { return (H == 0 ? XBUF1 : XBUF0) - priv<H>(); }
};

template <int CODE, class B, int HALF>
struct SplitGroup {
    using V = typename B::V;
    using GEO = Geo<CODE, HALF>;
    using LAY = SplitLayout<CODE>;
    using A = Arith<B>;
    static constexpr int M = GEO::M, N = GEO::N, L = GEO::L, W = GEO::W, G = GEO::G, NTX = GEO::NTX, NCOLS = GEO::NCOLS, NROWS = GEO::NROWS, Q = GEO::Q,
                         ARG = GEO::ARG;
    static constexpr int XO = LAY::template own_x<HALF>(), XT = LAY::template other_x<HALF>();
    template <int TT> static BS_FN V op3(V a, V b, V c) { return B::template bitop3<TT>(a, b, c); }

    Decoder<CODE, B, HALF> d;
    V lane, cw, lw, frame, valid, extra_fail;
    uint64_t valid_mask = 0, frozen_mask = 0, ok_mask = 0;
    uint32_t iters_s[G];                       // verdicts are wave-uniform per codeword: scalars, not planes
    const int8_t *llrs = nullptr;
    uint8_t *output = nullptr;
    uint32_t *iters = nullptr;
    uint8_t *success = nullptr;

    // once per kernel and wave: the lane permutation tables of the owned exchanged edges
    BS_FN void init(B &b) { d.init_lane(b); d.init_perm_tables(b); }

    // ---- LLRs of the owned transmitted columns -> bit planes in LDS; state zeroed ----
    BS_FN void prologue(B &b, const int8_t *llrs_all, uint8_t *output_all, uint32_t *iters_all, uint8_t *success_all, uint32_t batch,
                        uint32_t maxiters, uint32_t group)
    {
        d.init_lane(b);
        lane = d.lane;
        cw = B::shr(lane, ilog2c(W));
        lw = B::and_(lane, B::c(W - 1));
        llrs = llrs_all + (size_t)group * G * GEO::N;
        output = output_all + (size_t)group * G * GEO::OUT_LEN;
        iters = iters_all + (size_t)group * G;
        success = success_all + (size_t)group * G;
        frame = cw;
        valid = B::less_u(B::add(B::c(group * (uint32_t)G), cw), B::c(batch));
        valid_mask = b.ballot(valid);
        sfor<0, NTX>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (GEO::owns_col(c)) {
                V X[8];
                load_column_planes<CODE>(b, d, llrs, c, group, batch, XO, X);      // (the staging slab is this wave's exchange buffer)
                sfor<0, 8>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    b.lds_write32(B::add(B::shl(lane, 2), B::c(GEO::LDS_LLR + (GEO::llr_slot(c) * LLRP + k) * 256)), X[k]);
                });
            }
        });
        d.reset_state(b);
        d.prime_perm(b);
        frozen_mask = ~valid_mask;
        sfor<0, G>([&](auto G_) { iters_s[decltype(G_)::value] = maxiters; });
        ok_mask = 0;
        extra_fail = B::c(0);
    }

    BS_FN bool running() const { return frozen_mask != ~0ull; }

    // ---- stage 1 of an iteration: the owned block columns ----
    BS_FN void stage_columns(B &b)
    {
        d.columns(b, frozen_mask);              // (the rows only this wave has edges in are finished in there: d.fail)
    }
    static BS_FN V xaddr(V lane, int base, int plane) { return B::add(B::shl(lane, 2), B::c(base + plane * 256)); }

    // The two shared block rows: this wave PUBLISHES its partial state of row PUB and later fetches that row's merged state; it MERGES
    // row MRG -- the one the other wave publishes -- and hands the result back.  One merge per wave, two barriers per iteration.
    static constexpr int nth_shared(int n) { for (int r = 0; r < NROWS; ++r) if (GEO::shared_row(r) && n-- == 0) return r; return -1; }
    static_assert(nth_shared(1) >= 0 && nth_shared(2) < 0, "exactly two block rows have edges in both halves");
    static constexpr int PUB = nth_shared(HALF), MRG = nth_shared(1 - HALF);

    template <int R> BS_FN void put_row(B &b, int base)
    {
        sfor<0, PL>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            b.lds_write32(xaddr(lane, base, k), d.W1[R][k]);
            b.lds_write32(xaddr(lane, base, PL + k), d.W2[R][k]);
        });
        b.lds_write32(xaddr(lane, base, 2 * PL), d.Sn[R]);
        b.lds_write32(xaddr(lane, base, 2 * PL + 1), d.Pn[R]);
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; b.lds_write32(xaddr(lane, base, 2 * PL + 2 + k), d.argn[R][k]); });
    }

    // ---- stage 2: the partial state of row PUB into this wave's buffer, with the parity of the rows only this wave has (the other
    // needs it for the verdict) ----
    BS_FN void stage_publish(B &b)
    {
        put_row<PUB>(b, XO);
        if constexpr (has_unshared(HALF)) b.lds_write32(xaddr(lane, XO, LAY::XEXTRA), d.fail);
    }
    // a half has block rows of its own (row 0 of the rate-4/5 codes is all wave 1's): their parity goes along for the other's verdict
    static constexpr bool has_unshared(int h) { for (int r = 0; r < NROWS; ++r) if (GEO::row_in_half(r, h) && !GEO::row_in_half(r, 1 - h)) return true; return false; }

    // ---- stage 3: the other wave's partial state of row MRG (in ITS buffer) merged into this wave's; the result goes back into that
    // buffer, where its owner fetches it ----
    BS_FN void stage_merge(B &b)
    {
        constexpr int R = MRG;
        V o1[PL], o2[PL], oarg[ARG > 0 ? ARG : 1];
        sfor<0, PL>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            o1[k] = b.lds_read32(xaddr(lane, XT, k));
            o2[k] = b.lds_read32(xaddr(lane, XT, PL + k));
        });
        const V os = b.lds_read32(xaddr(lane, XT, 2 * PL)), op = b.lds_read32(xaddr(lane, XT, 2 * PL + 1));
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; oarg[k] = b.lds_read32(xaddr(lane, XT, 2 * PL + 2 + k)); });
        if constexpr (has_unshared(1 - HALF)) extra_fail = b.lds_read32(xaddr(lane, XT, LAY::XEXTRA));
        // take the other's min1 where it is smaller -- wave 0's wins a tie, whichever wave merges
        V take;
        if constexpr (HALF == 0) take = A::less_than(o1, d.W1[R]);                  // other (wave 1) strictly smaller
        else take = B::not_(A::less_than(d.W1[R], o1));                             // mine (wave 1) not strictly smaller
        // the union's second smallest: the loser of the min1 comparison against the WINNER's side's min2 (the loser's own min2 is
        // no smaller than the loser) -- one more compare, not two
        V hi[PL], lo[PL];
        sfor<0, PL>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            hi[k] = op3<TT_MUX>(take, d.W1[R][k], o1[k]);                            // the larger of the two min1
            lo[k] = op3<TT_MUX>(take, o2[k], d.W2[R][k]);                            // min2 of the side whose min1 won
            d.W1[R][k] = op3<TT_MUX>(take, o1[k], d.W1[R][k]);
        });
        const V lt3 = A::less_than(hi, lo);
        sfor<0, PL>([&](auto K_) { constexpr int k = decltype(K_)::value; d.W2[R][k] = op3<TT_MUX>(lt3, hi[k], lo[k]); });
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; d.argn[R][k] = op3<TT_MUX>(take, oarg[k], d.argn[R][k]); });
        d.Sn[R] = B::xor_(d.Sn[R], os);
        d.Pn[R] = B::xor_(d.Pn[R], op);
        // the merged row is FINISHED here (minima back to magnitudes, new state -> old state) and handed back in that form: the
        // other wave takes 2 * 7 + 2 + ARG planes as they are instead of 2 * 8 + 2 + ARG and the conversion once more
        d.template finish_row<R>();
        sfor<0, MG>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            b.lds_write32(xaddr(lane, XT, k), d.m1[R][k]);
            b.lds_write32(xaddr(lane, XT, MG + k), d.m2[R][k]);
        });
        b.lds_write32(xaddr(lane, XT, 2 * MG), d.S[R]);
        b.lds_write32(xaddr(lane, XT, 2 * MG + 1), d.Pn[R]);
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; b.lds_write32(xaddr(lane, XT, 2 * MG + 2 + k), d.arg[R][k]); });
    }

    // ---- stage 4: the merged and finished state of row PUB, which the other wave left in this wave's buffer ----
    BS_FN void stage_fetch(B &b)
    {
        constexpr int R = PUB;
        sfor<0, MG>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            d.m1[R][k] = b.lds_read32(xaddr(lane, XO, k));
            d.m2[R][k] = b.lds_read32(xaddr(lane, XO, MG + k));
        });
        d.S[R] = b.lds_read32(xaddr(lane, XO, 2 * MG));
        d.fail = B::or_(d.fail, b.lds_read32(xaddr(lane, XO, 2 * MG + 1)));              // its parity: unsatisfied checks (:453)
        sfor<0, ARG>([&](auto K_) { constexpr int k = decltype(K_)::value; d.arg[R][k] = b.lds_read32(xaddr(lane, XO, 2 * MG + 2 + k)); });
    }

    // ---- stage 5: the verdict of iteration `it` (the same in both waves) ----
    BS_FN void stage_finish(B &b, uint32_t it)
    {
        const V fail = B::or_(d.fail, extra_fail);
        const uint64_t unsat_lanes = b.ballot(fail);
        sfor<0, G>([&](auto G_) {
            constexpr int g = decltype(G_)::value;
            constexpr uint64_t gm = (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1)) << ((g * W) & 63);
            if (!(unsat_lanes & gm) && !(frozen_mask & gm)) {                  // satisfied for the first time: (true, it)
                iters_s[g] = it;
                ok_mask |= gm;
                frozen_mask |= gm;
            }
        });
    }

    // ---- hard decisions of the owned columns, MSB first; iterations and success by wave 0 ----
    BS_FN void epilogue(B &b)
    {
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (GEO::owns_col(c)) b.lds_write32(B::add(B::shl(lane, 2), B::c(GEO::LDS_HARD + c_slot(c) * 256)), d.hard[c]);
        });
        sfor<0, NCOLS>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (GEO::owns_col(c)) {
                const V out = pack_hard_column<CODE>(b, d, GEO::LDS_HARD + c_slot(c) * 256);
                b.gstore32(output, B::add(B::mul_u(frame, (uint32_t)GEO::OUT_LEN), B::add(B::c((uint32_t)c * (M / 8)), B::shl(lw, 2))), out, valid);
            }
        });
        if constexpr (HALF == 0) {
            V iters_v = B::c(iters_s[0]);
            sfor<1, G>([&](auto G_) {
                constexpr int g = decltype(G_)::value;
                constexpr uint64_t gm = ((1ull << (W & 63)) - 1) << ((g * W) & 63);
                iters_v = B::select_lanes(gm, B::c(iters_s[g]), iters_v);
            });
            const V first = B::and_(valid, B::eq(lw, B::c(0)));
            b.gstore32(iters, B::shl(frame, 2), iters_v, first);
            b.gstore8(success, frame, B::and_(b.plane_of(ok_mask), B::c(1)), first);
        }
    }
    // ordinal of block column c among the owned ones (the epilogue's hard-decision words alias the LLR planes)
    static constexpr int c_slot(int c) { int s = 0; for (int i = 0; i < c; ++i) s += GEO::owns_col(i) ? 1 : 0; return s; }

    // ---- slot refill (round 6; decode_refill of decode_ms_bitslice.hpp for a group shared by two waves) --------------------------------
    // A slot whose codeword is finished hands in its results and takes the workgroup's next frame while the other slots iterate.  Both
    // waves run the SAME bookkeeping on the same verdicts (stage_finish: identical in both), so they agree on every slot and frame
    // without talking; each emits the output dwords of ITS block columns and loads the LLRs of ITS transmitted columns, per slot and
    // cooperatively (one (column, lane) unit per lane).  Sixteen slots (TM1280) would cost 32 scalar registers and a 16-way unrolled
    // event as scalars: iteration counts and frames live in two vector registers (a lane holds its slot's value), verdicts are lane
    // masks spread over a slot's lanes by scalar shifts, and an event walks the finished slots in a run-time loop (v_readlane for the
    // slot's frame).  Frames come in CHUNKS [lo, hi) from `draw()`, which hands BOTH waves the same sequence (the kernel: wave 0 draws
    // from the launch's queue word and passes the ticket on through LDS behind a barrier; both waves ask at the same points).
    // Scratch in LDS: this wave's exchange buffer, idle between stage_fetch and the next stage_publish: [hard-decision words of the
    // owned columns | staging slab].
    static constexpr int NCOLS_OWN = GEO::NCOLS_OWN, NTX_OWN = GEO::NTX_OWN, N_OWN = NTX_OWN * M;
    static constexpr int RF_HARD = XO, RF_STAGE = XO + NCOLS_OWN * 256;
    static_assert(NCOLS_OWN <= 8 && NCOLS_OWN * 256 + N_OWN + 16 * (N_OWN / 256) <= LAY::XBYTES, "the refill's scratch fits the exchange buffer");
    // the i-th owned block column / owned transmitted column, as packed 4-bit literals
    static constexpr uint32_t own_cols_lit(bool tx_only)
    {
        uint32_t lit = 0;
        int i = 0;
        for (int c = 0; c < NCOLS; ++c)
            if (GEO::owns_col(c) && (!tx_only || c < NTX)) lit |= (uint32_t)c << (4 * i++);
        return lit;
    }
    static constexpr uint64_t all_lanes_of(int) { return W == 64 ? ~0ull : ((1ull << (W & 63)) - 1); }
    V it_v, fr_v;                          // per lane: iterations done by / frame of the lane's slot
    uint64_t rf_active = 0, rf_fin = ~0ull, rf_ok = 0;
    uint32_t rf_next = 0, rf_end = 0;      // the current chunk's next frame without a slot, its end
    bool rf_supply = true;                 // draw() has not said "no more" yet
    static constexpr int RF_ROUNDS = (N_OWN + 1023) / 1024;
    V rf_ahead[RF_ROUNDS][4];              // the LLR bytes (owned columns) of the next frame, requested one frame ahead of their use
    uint32_t rf_ahead_frame = NO_FRAME;

    // the next frame of the supply (a new chunk from draw() when the current one is used up), or NO_FRAME
    template <class DRAW> BS_FN uint32_t rf_take(DRAW draw)
    {
        if (rf_next == rf_end) {
            if (!rf_supply) return NO_FRAME;
            const uint64_t chunk = draw();
            if (chunk == 0) { rf_supply = false; return NO_FRAME; }
            rf_next = (uint32_t)chunk;
            rf_end = (uint32_t)(chunk >> 32);
        }
        return rf_next++;
    }
    template <class DRAW> BS_FN void rf_look_ahead(B &b, DRAW draw)
    {
        rf_ahead_frame = rf_take(draw);
        if (rf_ahead_frame != NO_FRAME) fetch_owned(b, llrs + (size_t)rf_ahead_frame * GEO::N);
    }

    // a lane mask in which every slot with a set lane has ALL its lanes set
    static BS_FN uint64_t spread_slots(uint64_t m)
    {
        if constexpr (W == 64) return m ? ~0ull : 0ull;
        else {
            constexpr uint64_t FIRST = []() constexpr { uint64_t f = 0; for (int g = 0; g < G; ++g) f |= 1ull << (g * W); return f; }();
            uint64_t t = m;
            for (int sh = 1; sh < W; sh <<= 1) t |= t >> sh;       // (the slot's first lane collects the OR of the slot's lanes ...)
            t &= FIRST;
            for (int sh = 1; sh < W; sh <<= 1) t |= t << sh;       // (... and hands it back to all of them)
            return t;
        }
    }

    template <class DRAW>
    BS_FN void refill_begin(B &b, const int8_t *llrs_all, uint8_t *output_all, uint32_t *iters_all, uint8_t *success_all, DRAW draw)
    {
        d.init_lane(b);
        lane = d.lane;
        lw = B::and_(lane, B::c(W - 1));
        llrs = llrs_all; output = output_all; iters = iters_all; success = success_all;
        d.reset_state(b);
        d.prime_perm(b);
        it_v = B::c(0); fr_v = B::c(0);
        rf_active = 0; rf_fin = ~0ull; rf_ok = 0;
        rf_next = 0; rf_end = 0; rf_supply = true;
        extra_fail = B::c(0);
        rf_look_ahead(b, draw);
    }

    // the verdict of the iteration just run (the same in both waves): converged slots finish with (true, their iteration index), the
    // others count on; slots at the iteration cap finish as they are (decoder.rs:453-474)
    BS_FN void refill_verdict(B &b, uint32_t maxiters)
    {
        const V fail = B::or_(d.fail, extra_fail);
        const uint64_t unsat = spread_slots(b.ballot(fail));
        const uint64_t won = rf_active & ~unsat;
        rf_fin |= won;
        rf_ok |= won;
        it_v = B::select_lanes(rf_active & unsat, B::add(it_v, B::c(1)), it_v);
        refill_expire(b, maxiters);
    }
    BS_FN void refill_expire(B &b, uint32_t maxiters)
    {
        rf_fin |= rf_active & b.ballot(B::not_(B::less_u(it_v, B::c(maxiters))));          // it >= max_iters (max_iters = 0: at once)
    }

    // ---- an EVENT: the finished slots hand in their results, free slots take the next frames.  draw() -> (lo | (uint64_t)hi << 32) of
    // the next chunk, or 0 when there is none left (it is then not asked again) ----
    template <class DRAW>
    BS_FN void refill_event(B &b, DRAW draw)
    {
        d.reinit_lane();
        lane = d.lane;
        lw = B::and_(lane, B::c(W - 1));
        uint64_t done = rf_fin & rf_active;
        if (done) {
            sfor<0, NCOLS>([&](auto C_) {
                constexpr int c = decltype(C_)::value;
                if constexpr (GEO::owns_col(c)) b.lds_write32(B::add(B::shl(lane, 2), B::c(RF_HARD + c_slot(c) * 256)), d.hard[c]);
            });
            B::mem_fence();
            while (done) {
                const int first = __builtin_ctzll(done);                                  // the slot's first lane
                const uint32_t f = b.readlane(fr_v, first), itv = b.readlane(it_v, first);
                emit_owned(b, first, output + (size_t)f * GEO::OUT_LEN, iters + f, success + f, itv, (uint32_t)((rf_ok >> first) & 1));
                done &= ~(all_lanes_of(0) << first);
            }
        }
        rf_active &= ~rf_fin;
        uint64_t vacant = rf_fin, fresh = 0;
        while (vacant && rf_ahead_frame != NO_FRAME) {
            const int first = __builtin_ctzll(vacant);
            const uint64_t gmask = all_lanes_of(0) << first;
            const uint32_t f = rf_ahead_frame;
            place_owned(b, first);
            rf_look_ahead(b, draw);
            fr_v = B::select_lanes(gmask, B::c(f), fr_v);
            it_v = B::select_lanes(gmask, B::c(0), it_v);
            fresh |= gmask;
            vacant &= ~gmask;
        }
        rf_active |= fresh;
        if (fresh) d.reset_state(b, fresh);
        rf_fin = 0; rf_ok = 0;
    }

    // the finished codeword of the slot whose first lane is `first`: the hard-decision words of the OWNED block columns -> their output
    // dwords, one (column, lane) unit per lane; wave 0 also stores the iteration count and the flag
    BS_FN void emit_owned(B &b, int first, uint8_t *out, uint32_t *iters_p, uint8_t *ok_p, uint32_t iters_v, uint32_t ok_v)
    {
        constexpr int UNITS = NCOLS_OWN * W;
        sfor<0, (UNITS + 63) / 64>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V u = B::add(lane, B::c(64 * r));
            const V valid = B::less_u(u, B::c(UNITS));
            const V i = B::and_(B::shr(u, ilog2c(W)), valid);                         // ordinal of the owned column
            const V c = B::bfe(B::c(own_cols_lit(false)), B::shl(i, 2), 4);            // the column itself
            const V ulw = B::and_(u, B::c(W - 1));
            const V b0 = B::shl(B::and_(ulw, B::c(L - 1)), 5 - ilog2c(L));
            const V qbase = B::add(B::shl(i, 8), B::shl(B::add(B::c(first), B::and_(ulw, B::c(~(L - 1) & (W - 1)))), 2));
            V o = B::c(0);
            sfor<0, L>([&](auto LL_) {
                constexpr int l2 = decltype(LL_)::value;
                const V w = B::shr_v(b.lds_read32(B::add(qbase, B::c(RF_HARD + 4 * l2))), b0);
                sfor<0, 32 / L>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    constexpr int t = l2 + L * k;
                    constexpr int pos = 8 * (t / 8) + 7 - (t % 8);
                    o = B::or_(o, B::shl(B::and_(B::shr(w, k), B::c(1)), pos));
                });
            });
            b.gstore32(out, B::add(B::mul_u(c, (uint32_t)(M / 8)), B::shl(ulw, 2)), o, valid);
        });
        if constexpr (HALF == 0) {
            const V first_lane = B::eq(lane, B::c(0));
            b.gstore32(iters_p, B::c(0), B::c(iters_v), first_lane);
            b.gstore8(ok_p, B::c(0), B::c(ok_v), first_lane);
        }
    }

    // position among the owned columns' bytes of the lane's 16 bytes of round R (the last round's idle lanes repeat the last 16)
    template <int R> BS_FN V rf_pos() const
    {
        V p = B::add(B::shl(lane, 4), B::c(1024 * R));
        if constexpr (1024 * (R + 1) > N_OWN) {
            const V over = B::less_u(B::c(N_OWN - 16), p);
            p = B::template bitop3<TT_MUX>(over, B::c(N_OWN - 16), p);
        }
        return p;
    }
    // the LLR bytes of the OWNED transmitted columns of a frame into rf_ahead (16 bytes per lane and round)
    BS_FN void fetch_owned(B &b, const int8_t *src)
    {
        sfor<0, RF_ROUNDS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V p = rf_pos<r>();
            const V c = B::bfe(B::c(own_cols_lit(true)), B::shl(B::shr(p, ilog2c(M)), 2), 4);
            b.gload128(src, B::add(B::shl(c, ilog2c(M)), B::and_(p, B::c(M - 1))), rf_ahead[r]);
        });
    }
    // a fresh codeword for the slot whose first lane is `first`: rf_ahead -> staging slab -> one (column, lane) unit of 32 LLRs per
    // lane -> its 8 bit planes
    BS_FN void place_owned(B &b, int first)
    {
        constexpr int UNITS = NTX_OWN * W;
        sfor<0, RF_ROUNDS>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V p = rf_pos<r>();
            b.lds_write128(B::add(B::add(p, B::shl(B::shr(p, 8), 4)), B::c(RF_STAGE)), rf_ahead[r]);
        });
        B::mem_fence();
        sfor<0, (UNITS + 63) / 64>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            const V u = B::add(lane, B::c(64 * r));
            const V valid = B::less_u(u, B::c(UNITS));
            const V i = B::and_(B::shr(u, ilog2c(W)), valid);
            const V ulw = B::and_(u, B::c(W - 1));
            const V p0 = B::add(B::add(B::shl(i, ilog2c(M)), B::shl(B::shr(ulw, ilog2c(L)), ilog2c(Q))), B::and_(ulw, B::c(L - 1)));
            V X[8];
            gather_planes<CODE>(b, p0, RF_STAGE, X);
            const V at = B::add(B::mul_u(i, (uint32_t)(LLRP * 256)), B::shl(B::add(B::c(first), ulw), 2));
            sfor<0, 8>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                b.lds_write32_if(B::add(at, B::c(GEO::LDS_LLR + k * 256)), X[k], valid);
            });
        });
        B::mem_fence();
    }
};

// One wave's program for a group: the HIP kernel calls it with a workgroup barrier for SYNC, every wave of the pair with its HALF.
template <int CODE, class B, int HALF, class SYNC>
BS_FN void decode_group_split(B &b, SplitGroup<CODE, B, HALF> &g, const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, uint32_t batch,
                              uint32_t maxiters, uint32_t group, SYNC sync)
{
    g.prologue(b, llrs, out, iters, ok, batch, maxiters, group);
    for (uint32_t it = 0; it < maxiters && g.running(); ++it) {
        g.stage_columns(b);
        g.stage_publish(b);                    // (own buffer: the other wave last touched it before the previous barrier)
        sync();                                // both partial states are in place
        g.stage_merge(b);                      // (the OTHER wave's buffer: read, then overwritten with the merged row)
        sync();                                // both merged rows are in place
        g.stage_fetch(b);                      // (own buffer)
        g.stage_finish(b, it);
    }
    sync();                                    // (the epilogue's hard-decision words alias nothing the other wave reads, but the next
    g.epilogue(b);                             //  group's staging slab is this buffer)
}

// One wave's program with slot refill: events between the iterations, the iteration itself is decode_group_split's (two barriers).
// Both waves take the same branches: every condition is a function of the shared verdicts and of draw()'s shared sequence.
template <int CODE, class B, int HALF, class SYNC, class DRAW>
BS_FN void decode_refill_split(B &b, SplitGroup<CODE, B, HALF> &g, const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, uint32_t maxiters,
                               SYNC sync, DRAW draw)
{
    g.refill_begin(b, llrs, out, iters, ok, draw);
    for (;;) {
        g.refill_event(b, draw);
        g.refill_expire(b, maxiters);                 // (max_iters = 0: the fresh slots are at the cap already)
        if (g.rf_fin) continue;
        if (!g.rf_active) break;
        do {
            g.d.columns(b, ~g.rf_active);
            g.stage_publish(b);
            sync();
            g.stage_merge(b);
            sync();
            g.stage_fetch(b);
            g.refill_verdict(b, maxiters);
        } while (!g.rf_fin);
    }
}

}  // namespace bs
}  // namespace ldpc
