// decode_ms_launch.hpp -- host-side dispatch from (code, variant) to a kernel instantiation.
//
// Each LLR type has its own translation unit (decode_ms_f32.hip, ...) so the instantiations
// compile in parallel; they all expand LDPC_DEFINE_LAUNCHER below.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

#include "decode_ms_kernel.hpp"
#include "decode_ms_pair.hpp"

namespace ldpc {

// Launch the decoder for `batch` frames on `stream`.  `variant` = 0 picks the tuned default
// IPT (indices per thread) for the code; a positive value requests that IPT explicitly and
// yields hipErrorInvalidConfiguration if it was not instantiated.
template <class T>
hipError_t launch_decode_ms(int code, int variant, const T *llrs, uint8_t *output, uint32_t *iters,
                            uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream);

// Largest |LLR| for which the f32 kernels may drop the FLT_MAX clamp of the exclusive minimum
// (decoder.rs:414-415) for a run of `maxiters` iterations.  The clamp acts only if a magnitude overflows to
// infinity.  With L = max |LLR|, V_k = max |v| and U_k = max |u| after iteration k:  U_k <= V_k (an exclusive
// minimum is one of the other |v|),  |va| <= L + 6 U_k (variable degree <= 6 in every code),
// |v| = |va - u| <= L + 7 U_k,  hence  V_k <= L (1 + 7 + ... + 7^k) < L 7^(k+1) / 6  and every sum stays below
// L * 7^maxiters * 1.4 (rounding included).  Requiring that to stay under 2^127 gives the limit below:
// 2^55 at the benchmark's 25 iterations, less than any real LLR from 45 iterations on (then every codeword
// simply takes the clamped copy of the loop).  0 = never.
inline float nocap_limit_for(uint32_t maxiters)
{
    const double log2_limit = 126.0 - 2.8074 * (double)maxiters;      // log2(7) = 2.80735...
    if (log2_limit < -120.0) return 0.0f;
    return (float)__builtin_ldexp(1.0, (int)__builtin_floor(log2_limit));
}

// Resident workgroups per device for one instantiation (occupancy x compute units), cached.
template <int CODE, class T, int IPT, bool PF, int LEAN>
int resident_workgroups()
{
    using GEO = Geometry<CODE, T, IPT>;
    static std::atomic<int> cached[64] = {};     // concurrent callers may both fill an entry: they store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_kernel<CODE, T, IPT, PF, LEAN>, GEO::WG, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        v = per_cu * cus;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Launch one instantiation (IPT indices per thread; LEAN 1 = register-lean check phase, 2 = in-place messages).
template <int CODE, class T, int IPT, int LEAN>
hipError_t launch_cfg(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                      size_t batch, uint32_t maxiters, hipStream_t stream)
{
    using GEO = Geometry<CODE, T, IPT>;
    // LLR staging (PF) is implemented but measured SLOWER than plain loads at the start of each
    // codeword on TM8192 (4.99 vs 5.29 M codewords/s: the extra live state costs spills at the
    // 128-VGPR budget), so it stays off; see DESIGN.md section 7.
    constexpr bool PF = false;
    if (batch == 0) return hipSuccess;
    const size_t groups = (batch + GEO::G - 1) / GEO::G;
    if (batch > 0xFFFFFFFFull || groups > 0x7FFFFFFFull) return hipErrorInvalidValue;   // (capi.hip slices larger batches)
    // Persistent workgroups with a static stride over the codeword groups.  Where one workgroup
    // fills a CU (TM8192) the grid is exactly the resident set: each workgroup then decodes
    // hundreds of codewords and the data-dependent iteration counts average out.  Where several
    // fit per CU the grid is 16x the resident set so that the hardware dispatcher still balances
    // (measured on TM2048: 21.7 / 25.0 / 27.2 / 27.7 M codewords/s at 1x / 2x / 8x / 64x).
    const size_t resident = (size_t)resident_workgroups<CODE, T, IPT, PF, LEAN>();
    size_t grid = resident <= 256 ? resident : resident * 16;
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL((decode_ms_kernel<CODE, T, IPT, PF, LEAN>), dim3((unsigned)grid), dim3(GEO::WG), 0, stream,
                       llrs, output, iters, success, (uint32_t)batch, maxiters, nocap_limit_for(maxiters));
    return hipGetLastError();
}

template <int CODE, class T, int IPT>
hipError_t launch_one(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                      size_t batch, uint32_t maxiters, hipStream_t stream)
{
    // Register-lean variant (decode_ms_kernel.hpp): pays where it doubles the workgroups per CU, which
    // is TM5120 (f32 12.4 -> 13.8, i8 11.3 -> 13.8 M codewords/s at 4 dB) and the narrow types of TM1280.  Measured slower
    // on TM1280 / TM1536 / TM2048 / TM6144 f32 (-7..-8 %), whose occupancy it does not change.
    // TM1280 i8 / i16: 168 -> 116 VGPRs, four waves per SIMD instead of three: 68.4 -> 71.0 (its f32 kernel 70.5 -> 69.5: not).
    constexpr bool narrow = std::is_same_v<T, int8_t> || std::is_same_v<T, int16_t>;
    constexpr int LEAN = (CODE == TM5120 || (CODE == TM1280 && narrow)) && IPT == 1 && !std::is_same_v<T, int32_t> ? 1 : 0;   // (i32's wider integer sequences spill at the lean kernel's 128-VGPR budget)
    return launch_cfg<CODE, T, IPT, LEAN>(llrs, output, iters, success, batch, maxiters, stream);
}

// Pair-ownership kernel (decode_ms_pair.hpp): one workgroup per CU-resident codeword, persistent.
constexpr int VARIANT_PAIR = 32;
template <int CODE, class T>
hipError_t launch_pair(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                       size_t batch, uint32_t maxiters, hipStream_t stream)
{
    using GEO = PairGeometry<CODE, T>;
    if (batch == 0) return hipSuccess;
    if (batch > 0x7FFFFFFFull) return hipErrorInvalidValue;
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int resident = cached[dev].load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_pair_kernel<CODE, T>, GEO::NT, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = per_cu * cus;
        cached[dev].store(resident, std::memory_order_relaxed);
    }
    size_t grid = resident <= 256 ? (size_t)resident : (size_t)resident * 16;
    if (grid > batch) grid = batch;
    hipLaunchKernelGGL((decode_ms_pair_kernel<CODE, T>), dim3((unsigned)grid), dim3(GEO::NT), 0, stream,
                       llrs, output, iters, success, (uint32_t)batch, maxiters, nocap_limit_for(maxiters));
    return hipGetLastError();
}

// one `case` of the dispatch switch: default IPT plus optional alternatives
#define LDPC_CASE(CODE, T, DEF, ...)                                                             \
    case CODE: {                                                                                 \
        constexpr int alts[] = {DEF, ##__VA_ARGS__};                                             \
        return dispatch_ipt<CODE, T, DEF, ##__VA_ARGS__>(variant == 0 ? alts[0] : variant, llrs, \
                                                         output, iters, success, batch, maxiters, \
                                                         stream);                                \
    }

template <int CODE, class T, int... IPTS>
hipError_t dispatch_ipt(int ipt, const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                        size_t batch, uint32_t maxiters, hipStream_t stream)
{
    hipError_t r = hipErrorInvalidConfiguration;
    (void)((ipt == IPTS ? (r = launch_one<CODE, T, IPTS>(llrs, output, iters, success, batch, maxiters, stream), true)
                        : false) || ...);
    return r;
}

}  // namespace ldpc
