// CPU emulation of the bit-sliced i8 decoder (labrador_ldpc_amd/csrc/decode_ms_bitslice.hpp): the SAME source text the
// gfx950 kernel is compiled from, instantiated with a backend whose "wave register" is an array of 64 lanes and whose
// cross-lane / LDS / memory operations are plain loops.  Test infrastructure (tests/test_bitslice_emu.py compares its
// results with the oracle on the CPU, so the formulation -- layout, lane permutations, plane arithmetic, compressed row
// state -- is validated without a GPU); it is not part of the product and nothing in labrador_ldpc_amd/ links it.
//   g++ -O1 -std=c++20 -shared -fPIC -DEMU_CODE=TM8192 -Ilabrador_ldpc_amd/csrc tests/c/bitslice_emu.cpp -o build/libbitslice_emu_TM8192.so
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define BS_FN inline
#include "decode_ms_bitslice.hpp"
#include "decode_ms_bitslice_split.hpp"
#include "decode_bf_bitslice.hpp"

namespace {

struct Vec {
    uint32_t l[64];
};

struct EmuBackend {
    using V = Vec;
    // the workgroup's LDS (bounds-checked) and this wave's base in it: the two waves of a split group share one store
    std::shared_ptr<std::vector<uint8_t>> store;
    size_t base = 0;
    explicit EmuBackend(size_t lds_bytes) : store(std::make_shared<std::vector<uint8_t>>(lds_bytes, 0xA5)) {}
    EmuBackend(std::shared_ptr<std::vector<uint8_t>> shared, size_t base_) : store(std::move(shared)), base(base_) {}
    struct Lds {
        std::vector<uint8_t> *v; size_t base;
        uint8_t &at(size_t a) const { return v->at(base + a); }
    };
    Lds lds_() const { return Lds{store.get(), base}; }

    template <class F> static V map1(const V &a, F f) { V r; for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i]); return r; }
    template <class F> static V map2(const V &a, const V &b, F f) { V r; for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i], b.l[i]); return r; }

    static void fence() {}
    static void lds_wait() {}
    static void mem_fence() {}
    static void pin(V &) {}
    static V c(uint32_t x) { V r; for (auto &e : r.l) e = x; return r; }
    V lane() const { V r; for (int i = 0; i < 64; ++i) r.l[i] = (uint32_t)i; return r; }
    template <int TT> static V bitop3(const V &a, const V &b, const V &cc)
    {
        V r;
        for (int i = 0; i < 64; ++i) {
            uint32_t o = 0;
            for (int m = 0; m < 8; ++m)
                if ((TT >> m) & 1) o |= ((m & 4) ? a.l[i] : ~a.l[i]) & ((m & 2) ? b.l[i] : ~b.l[i]) & ((m & 1) ? cc.l[i] : ~cc.l[i]);
            r.l[i] = o;
        }
        return r;
    }
    static V and_(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x & y; }); }
    static V or_(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x | y; }); }
    static V xor_(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x ^ y; }); }
    static V andn(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x & ~y; }); }
    static V not_(const V &a) { return map1(a, [](uint32_t x) { return ~x; }); }
    static V add(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x + y; }); }
    static V sub(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x - y; }); }
    static V mul_u(const V &a, uint32_t k) { return map1(a, [k](uint32_t x) { return x * k; }); }
    static V shl(const V &a, int s) { return map1(a, [s](uint32_t x) { return x << s; }); }
    static V shr(const V &a, int s) { return map1(a, [s](uint32_t x) { return x >> s; }); }
    static V sar(const V &a, int s) { return map1(a, [s](uint32_t x) { return (uint32_t)((int32_t)x >> s); }); }
    static V shl_v(const V &a, const V &s) { return map2(a, s, [](uint32_t x, uint32_t y) { return x << (y & 31); }); }
    static V shr_v(const V &a, const V &s) { return map2(a, s, [](uint32_t x, uint32_t y) { return x >> (y & 31); }); }
    static V bfe(const V &v, const V &off, int width) { return map2(v, off, [width](uint32_t x, uint32_t o) { return (x >> (o & 31)) & ((1u << width) - 1); }); }
    static V rotr(const V &x, const V &amt) { return map2(x, amt, [](uint32_t v, uint32_t a) { a &= 31; return a ? (v >> a) | (v << (32 - a)) : v; }); }
    static V less_u(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x < y ? 0xFFFFFFFFu : 0u; }); }
    static V eq(const V &a, const V &b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x == y ? 0xFFFFFFFFu : 0u; }); }
    template <int CTRL> static V quad_perm(const V &x) { V r; for (int i = 0; i < 64; ++i) r.l[i] = x.l[(i & ~3) | ((CTRL >> (2 * (i & 3))) & 3)]; return r; }
    V bperm(const V &addr, const V &x) const { V r; for (int i = 0; i < 64; ++i) r.l[i] = x.l[(addr.l[i] >> 2) & 63]; return r; }
    V lds_read32(const V &addr) const { V r; for (int i = 0; i < 64; ++i) std::memcpy(&r.l[i], &lds_().at(addr.l[i]), 4), (void)lds_().at(addr.l[i] + 3); return r; }
    void lds_write32(const V &addr, const V &v) { for (int i = 0; i < 64; ++i) { (void)lds_().at(addr.l[i] + 3); std::memcpy(&lds_().at(addr.l[i]), &v.l[i], 4); } }
    V lds_read_u16(const V &addr) const { V r; for (int i = 0; i < 64; ++i) r.l[i] = lds_().at(addr.l[i]) | (uint32_t)lds_().at(addr.l[i] + 1) << 8; return r; }
    void lds_write16(const V &addr, const V &v) { for (int i = 0; i < 64; ++i) { lds_().at(addr.l[i]) = (uint8_t)v.l[i]; lds_().at(addr.l[i] + 1) = (uint8_t)(v.l[i] >> 8); } }
    void lds_write32_if(const V &addr, const V &v, const V &pred) { for (int i = 0; i < 64; ++i) if (pred.l[i]) { (void)lds_().at(addr.l[i] + 3); std::memcpy(&lds_().at(addr.l[i]), &v.l[i], 4); } }
    V lds_read_u8(const V &addr) const { V r; for (int i = 0; i < 64; ++i) r.l[i] = lds_().at(addr.l[i]); return r; }
    static V gload32(const void *p, const V &off, const V &pred)
    {
        V r;
        for (int i = 0; i < 64; ++i) { r.l[i] = 0; if (pred.l[i]) std::memcpy(&r.l[i], (const char *)p + off.l[i], 4); }
        return r;
    }
    static V gload32(const void *p, const V &off) { V r; for (int i = 0; i < 64; ++i) std::memcpy(&r.l[i], (const char *)p + off.l[i], 4); return r; }
    static void gload128(const void *p, const V &off, V (&w)[4]) { for (int i = 0; i < 64; ++i) for (int k = 0; k < 4; ++k) std::memcpy(&w[k].l[i], (const char *)p + off.l[i] + 4 * k, 4); }
    void lds_write128(const V &addr, const V (&w)[4])
    {
        for (int i = 0; i < 64; ++i) {
            if (addr.l[i] % 16) throw std::runtime_error("ds_write_b128 at an address that is not 16-byte aligned");
            for (int k = 0; k < 4; ++k) { (void)lds_().at(addr.l[i] + 4 * k + 3); std::memcpy(&lds_().at(addr.l[i] + 4 * k), &w[k].l[i], 4); }
        }
    }
    static void gstore32(void *p, const V &off, const V &v, const V &pred) { for (int i = 0; i < 64; ++i) if (pred.l[i]) std::memcpy((char *)p + off.l[i], &v.l[i], 4); }
    static void gstore32_stream(void *p, const V &off, const V &v, const V &pred) { for (int i = 0; i < 64; ++i) if (pred.l[i]) std::memcpy((char *)p + off.l[i], &v.l[i], 4); }
    static void gstore8(void *p, const V &off, const V &v, const V &pred) { for (int i = 0; i < 64; ++i) if (pred.l[i]) ((uint8_t *)p)[off.l[i]] = (uint8_t)v.l[i]; }
    static V select_lanes(uint64_t m, const V &a, const V &b) { V r; for (int i = 0; i < 64; ++i) r.l[i] = ((m >> i) & 1) ? a.l[i] : b.l[i]; return r; }
    static uint32_t readlane(const V &x, int l) { return x.l[l & 63]; }
    static uint64_t ballot(const V &x) { uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (uint64_t)(x.l[i] != 0) << i; return m; }
    static V plane_of(uint64_t m) { V r; for (int i = 0; i < 64; ++i) r.l[i] = ((m >> i) & 1) ? 0xFFFFFFFFu : 0u; return r; }
};

template <int CODE>
int run(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    using GEO = ldpc::bs::Geo<CODE>;
    const size_t groups = (batch + GEO::G - 1) / GEO::G;
    for (size_t g = 0; g < groups; ++g) {
        EmuBackend b(GEO::LDS_BYTES);
        ldpc::bs::init_kernel<CODE, EmuBackend>(b);
        ldpc::bs::decode_group<CODE, EmuBackend>(b, llrs, out, iters, ok, (uint32_t)batch, maxiters, (uint32_t)g);
    }
    return 0;
}

// slot refill (decode_refill): ONE emulated wave takes the whole batch, frame after frame, a finished slot taking the next frame
template <int CODE>
int run_refill(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    using GEO = ldpc::bs::Geo<CODE>;
    if constexpr (GEO::SPLIT || GEO::G < 2 || GEO::TWO_WAVES) return -1;
    else {
        EmuBackend b(GEO::LDS_BYTES);
        ldpc::bs::init_kernel<CODE, EmuBackend>(b);
        uint32_t nextf = 0;
        ldpc::bs::decode_refill<CODE, EmuBackend>(b, llrs, out, iters, ok, maxiters,
                                                  [&]() -> uint32_t { return nextf < batch ? nextf++ : ldpc::bs::NO_FRAME; });
        return 0;
    }
}

// the two-waves-per-group kernel of the rate-4/5 codes (decode_ms_bitslice_split.hpp): the two halves of a group run stage by stage,
// alternately, on one shared LDS store -- the order the workgroup barriers enforce on the GPU
template <int CODE>
int run_split(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    using namespace ldpc::bs;
    using LAY = SplitLayout<CODE>;
    using GEO = Geo<CODE, 0>;
    const size_t groups = (batch + GEO::G - 1) / GEO::G;
    for (size_t g = 0; g < groups; ++g) {
        auto store = std::make_shared<std::vector<uint8_t>>(LAY::BYTES, 0xA5);
        EmuBackend b0(store, LAY::PRIV0), b1(store, LAY::PRIV1);
        SplitGroup<CODE, EmuBackend, 0> w0;
        SplitGroup<CODE, EmuBackend, 1> w1;
        w0.init(b0); w1.init(b1);
        w0.prologue(b0, llrs, out, iters, ok, (uint32_t)batch, maxiters, (uint32_t)g);
        w1.prologue(b1, llrs, out, iters, ok, (uint32_t)batch, maxiters, (uint32_t)g);
        for (uint32_t it = 0; it < maxiters && w0.running(); ++it) {
            if (w0.running() != w1.running()) return 2;                  // the two waves must agree on every verdict
            w0.stage_columns(b0); w1.stage_columns(b1);
            w0.stage_publish(b0); w1.stage_publish(b1);
            w0.stage_merge(b0); w1.stage_merge(b1);
            w0.stage_fetch(b0); w1.stage_fetch(b1);
            w0.stage_finish(b0, it); w1.stage_finish(b1, it);
            if (w0.frozen_mask != w1.frozen_mask) return 3;
        }
        w0.epilogue(b0); w1.epilogue(b1);
    }
    return 0;
}

// slot refill on the two-wave kernel (decode_range_split): ONE emulated workgroup takes the whole batch as its range; the two halves
// run stage by stage on one LDS store, as run_split does
template <int CODE>
int run_split_refill(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    using namespace ldpc::bs;
    using LAY = SplitLayout<CODE>;
    auto store = std::make_shared<std::vector<uint8_t>>(LAY::BYTES, 0xA5);
    EmuBackend b0(store, LAY::PRIV0), b1(store, LAY::PRIV1);
    SplitGroup<CODE, EmuBackend, 0> w0;
    SplitGroup<CODE, EmuBackend, 1> w1;
    w0.init(b0); w1.init(b1);
    // chunks of 5 frames (not a multiple of any G): each wave walks the same sequence with a cursor of its own
    uint32_t cur0 = 0, cur1 = 0;
    auto draw_from = [&](uint32_t &cur) { return [&cur, batch]() -> uint64_t {
        if (cur >= batch) return 0;
        const uint32_t lo = cur, hi = (uint32_t)(lo + 5 < batch ? lo + 5 : batch);
        cur = hi;
        return (uint64_t)lo | (uint64_t)hi << 32; }; };
    w0.refill_begin(b0, llrs, out, iters, ok, draw_from(cur0));
    w1.refill_begin(b1, llrs, out, iters, ok, draw_from(cur1));
    for (;;) {
        w0.refill_event(b0, draw_from(cur0)); w1.refill_event(b1, draw_from(cur1));
        w0.refill_expire(b0, maxiters); w1.refill_expire(b1, maxiters);
        if (w0.rf_fin != w1.rf_fin || w0.rf_active != w1.rf_active) return 2;        // the two waves must agree on every slot
        if (w0.rf_fin) continue;
        if (!w0.rf_active) break;
        do {
            w0.d.columns(b0, ~w0.rf_active); w1.d.columns(b1, ~w1.rf_active);
            w0.stage_publish(b0); w1.stage_publish(b1);
            w0.stage_merge(b0); w1.stage_merge(b1);
            w0.stage_fetch(b0); w1.stage_fetch(b1);
            w0.refill_verdict(b0, maxiters); w1.refill_verdict(b1, maxiters);
            if (w0.rf_fin != w1.rf_fin || w0.rf_ok != w1.rf_ok) return 3;
        } while (!w0.rf_fin);
    }
    return 0;
}

}  // namespace

// One code per shared object (-DEMU_CODE=TM8192 ...): the six instantiations compile in parallel (tests/test_bitslice_emu.py).
#ifndef EMU_CODE
#error "compile with -DEMU_CODE=<TM1280|TM1536|TM2048|TM5120|TM6144|TM8192>"
#endif
extern "C" int bs_emu_decode(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    // (the rate-4/5 codes exist only in the two-waves-per-group form)
    if constexpr (ldpc::bs::Geo<ldpc::EMU_CODE>::TWO_WAVES) return run_split<ldpc::EMU_CODE>(llrs, out, iters, ok, batch, maxiters);
    else return run<ldpc::EMU_CODE>(llrs, out, iters, ok, batch, maxiters);
}
extern "C" int bs_emu_decode_bf(const uint8_t *input, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    using GEO = ldpc::bs::Geo<ldpc::EMU_CODE>;
    const size_t groups = (batch + GEO::G - 1) / GEO::G;
    for (size_t g = 0; g < groups; ++g) {
        EmuBackend b(ldpc::bs::BfGeo<ldpc::EMU_CODE>::LDS_BYTES);
        ldpc::bs::bf_init_kernel<ldpc::EMU_CODE, EmuBackend>(b);
        ldpc::bs::bf_decode_group<ldpc::EMU_CODE, EmuBackend>(b, input, out, iters, ok, (uint32_t)batch, maxiters, (uint32_t)g);
    }
    return 0;
}
extern "C" int bs_emu_decode_split(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    if constexpr (ldpc::bs::Geo<ldpc::EMU_CODE>::TWO_WAVES) return run_split<ldpc::EMU_CODE>(llrs, out, iters, ok, batch, maxiters);
    else return -1;                                                     // rate 4/5 only
}
extern "C" int bs_emu_decode_refill(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    return run_refill<ldpc::EMU_CODE>(llrs, out, iters, ok, batch, maxiters);
}
extern "C" int bs_emu_decode_split_refill(const int8_t *llrs, uint8_t *out, uint32_t *iters, uint8_t *ok, size_t batch, uint32_t maxiters)
{
    if constexpr (ldpc::bs::Geo<ldpc::EMU_CODE>::TWO_WAVES) return run_split_refill<ldpc::EMU_CODE>(llrs, out, iters, ok, batch, maxiters);
    else return -1;
}
extern "C" int bs_emu_group(void) { return ldpc::bs::Geo<ldpc::EMU_CODE>::G; }
extern "C" int bs_emu_code(void) { return ldpc::EMU_CODE; }
