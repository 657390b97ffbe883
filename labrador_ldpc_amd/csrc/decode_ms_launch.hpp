// decode_ms_launch.hpp -- host-side dispatch from (code, variant) to a kernel instantiation.
//
// Each LLR type has its own translation unit (decode_ms_f32.hip, ...) so the instantiations
// compile in parallel; they all expand LDPC_DEFINE_LAUNCHER below.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "decode_ms_kernel.hpp"
#include "decode_ms_pair.hpp"

namespace ldpc {

// Launch the decoder for `batch` frames on `stream`.  `variant` = 0 picks the tuned default
// IPT (indices per thread) for the code; a positive value requests that IPT explicitly and
// yields hipErrorInvalidConfiguration if it was not instantiated.  VARIANT_STATIC added to either
// distributes the codewords over the workgroups by a fixed stride instead of through the launch's queue.
template <class T>
hipError_t launch_decode_ms(int code, int variant, const T *llrs, uint8_t *output, uint32_t *iters,
                            uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream);

// Will launch_decode_ms<T>(code, variant, ...) read every LLR from memory exactly once?  (false for explicit variants, which
// are not second-guessed, and for forced two-pass NaN handling, whose second kernel reads the first one's marks.)
template <class T>
bool decode_ms_reads_llrs_once(int code, int variant);

// Largest |LLR| for which the f32 kernels may drop the FLT_MAX clamp of the exclusive minimum
// (decoder.rs:414-415) for a run of `maxiters` iterations.  The clamp acts only if a magnitude overflows to
// infinity.  With L = max |LLR|, V_k = max |v| and U_k = max |u| after iteration k:  U_k <= V_k (an exclusive
// minimum is one of the other |v|),  |va| <= L + 6 U_k (variable degree <= 6 in every code),
// |v| = |va - u| <= L + 7 U_k,  hence  V_k <= L (1 + 7 + ... + 7^k) < L 7^(k+1) / 6  and every sum stays below
// L * 7^maxiters * 1.4 (rounding included).  Requiring that to stay under 2^127 gives the limit below:
// 2^55 at the benchmark's 25 iterations, less than any real LLR from 45 iterations on (then every codeword
// simply takes the clamped copy of the loop).  0 = never.
//
// `clamp_form`: the kernel runs the self-correction of bounded codewords as v = med3(nv, 0, nv + old * 2^126)
// (Ops<float>::clamp_to_side, form 2), which additionally needs 2^126 * |old| > |nv| for every nonzero old: with every
// nonzero value of the decode >= 2^-43 (the vote's lower bound 2^-20 on nonzero |LLR|, times the 2^-23 granularity)
// that is 2^83 > 1.4 L 7^maxiters, i.e. log2 L < 82.5 - log2(7) maxiters: 2^12 at the benchmark's 25 iterations.
inline float nocap_limit_for(uint32_t maxiters, bool clamp_form = false)
{
    double log2_limit = 126.0 - 2.8074 * (double)maxiters;            // log2(7) = 2.80735...
    if (clamp_form && 82.5 - 2.8074 * (double)maxiters < log2_limit) log2_limit = 82.5 - 2.8074 * (double)maxiters;
    if (log2_limit < -20.0) return 0.0f;                               // (below the vote's lower bound: no codeword passes)
    return (float)__builtin_ldexp(1.0, (int)__builtin_floor(log2_limit));
}

// `variant` flags; the rest of `variant` selects the kernel.  VARIANT_STATIC: fixed-stride distribution of the codewords (no
// queue).  VARIANT_ONE_PASS / VARIANT_TWO_PASS: force the one-kernel / the two-kernel handling of NaN LLRs for the kernels
// that have both (two_pass_nan() below; by default the batch size decides) -- for tests and A/B timing.
constexpr int VARIANT_STATIC = 256, VARIANT_ONE_PASS = 512, VARIANT_TWO_PASS = 1024;
constexpr int VARIANT_FLAGS = VARIANT_STATIC | VARIANT_ONE_PASS | VARIANT_TWO_PASS;
// the same as launch flags (`lflags`), split off `variant` by LDPC_SPLIT_VARIANT
constexpr unsigned LF_STATIC = 1, LF_ONE_PASS = 2, LF_TWO_PASS = 4;

// The queue head of a launch's dynamic codeword distribution (decode_ms_body, "dynamic distribution"): a device word that
// is zero between launches -- the kernel that drew from it puts it back.  One word per (device, stream): launches of a
// stream run in order, so they can share it; launches of different streams may overlap, so they must not.  nullptr
// (= fixed stride) while `stream` is being captured into a graph (a graph can be replayed on any stream, next to
// anything), for the per-thread default stream (one handle, many streams), when LABRADOR_LDPC_HIP_STATIC is set, or if
// the word cannot be allocated.
inline uint32_t *claim_counter(hipStream_t stream)
{
    static const bool off = [] { const char *e = std::getenv("LABRADOR_LDPC_HIP_STATIC"); return e && *e && *e != '0'; }();
    if (off || stream == hipStreamPerThread) return nullptr;
    if (stream != nullptr) {                     // (the legacy stream cannot be captured; asking would disturb a global-mode capture)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    struct Block { uint32_t *base = nullptr; unsigned used = 0; uint32_t *spare = nullptr; };     // spare: a zeroed block that lost an allocation race
    constexpr unsigned PER_BLOCK = 1024;
    constexpr size_t MAX_HEADS = 8 * PER_BLOCK;      // streams beyond that (handle churn) run the fixed stride: the map stays bounded
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, uint32_t *> heads;
    static std::map<int, Block> blocks;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = heads.find({dev, stream});
        if (it != heads.end()) return it->second;
        if (heads.size() >= MAX_HEADS) return nullptr;
        Block &b = blocks[dev];
        if ((b.base == nullptr || b.used >= PER_BLOCK) && b.spare != nullptr) { b.base = b.spare; b.spare = nullptr; b.used = 0; }
        if (b.base != nullptr && b.used < PER_BLOCK) {
            uint32_t *head = b.base + b.used++;
            heads[{dev, stream}] = head;
            return head;
        }
    }
    // A new block of queue heads for this device, allocated and zeroed OUTSIDE the lock (other workers' launches go on) and
    // without a device-wide synchronisation: the zeroing runs on a private non-blocking stream and only that stream is waited
    // for.  hipMalloc / hipStreamCreate are not capture-safe, so the thread's capture mode is relaxed around them: another
    // thread's global-mode capture is not invalidated by this one's first use of a device (round 3 advice).
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    uint32_t *p = nullptr;
    hipStream_t zs = nullptr;
    bool ok = hipMalloc(&p, PER_BLOCK * sizeof(uint32_t)) == hipSuccess;
    if (ok) ok = hipStreamCreateWithFlags(&zs, hipStreamNonBlocking) == hipSuccess;
    if (ok) ok = hipMemsetAsync(p, 0, PER_BLOCK * sizeof(uint32_t), zs) == hipSuccess && hipStreamSynchronize(zs) == hipSuccess;
    if (zs) (void)hipStreamDestroy(zs);
    if (!ok) { (void)hipGetLastError(); if (p) (void)hipFree(p); p = nullptr; }
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (p == nullptr) return nullptr;
    // (Nothing is freed under the lock: hipFree synchronises the whole device and is not capture-safe -- under the lock it would stall
    // every other worker's launch and could invalidate another thread's global-mode capture.  A block that lost the race below is
    // kept as the device's spare and becomes its next block; a second loser of the same race -- two spares at once -- is leaked,
    // 4 KB, once.)
    std::lock_guard<std::mutex> lock(mu);
    Block &b = blocks[dev];
    auto it = heads.find({dev, stream});             // (another thread may have served this stream meanwhile)
    if (it != heads.end()) { if (b.spare == nullptr) b.spare = p; return it->second; }
    if (b.base != nullptr && b.used < PER_BLOCK) { if (b.spare == nullptr) b.spare = p; }      // ... or installed a fresh block: use that one
    else { b.base = p; b.used = 0; }
    uint32_t *head = b.base + b.used++;
    heads[{dev, stream}] = head;
    return head;
}

// Resident workgroups per device for one instantiation (occupancy x compute units), cached.
template <int CODE, class T, int IPT, bool PF, int LEAN, int FORM, int NANPASS = 0>
int resident_workgroups()
{
    using GEO = Geometry<CODE, T, IPT>;
    static std::atomic<int> cached[64] = {};     // concurrent callers may both fill an entry: they store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_kernel<CODE, T, IPT, PF, LEAN, FORM, NANPASS>, GEO::WG, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        v = per_cu * cus;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Launch one instantiation (IPT indices per thread; LEAN 1 = register-lean check phase, 2 = in-place messages).
template <int CODE, class T, int IPT, int LEAN, int FORM, int NANPASS = 0>
hipError_t launch_cfg_form(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                           size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    const bool static_stride = (lflags & LF_STATIC) != 0 || NANPASS == 2;     // (the second NaN pass walks its own stride classes)
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
    // With the launch's queue (claim_counter) the grid is the resident set itself: the workgroups balance the work by
    // drawing from the queue, and nothing is paid for starting workgroups beyond the first wave of them.
    // Which distribution: the queue for workgroups of 8 waves and more (TM2048 +3.0 %, TM5120 +1.4-1.8 %, TM8192 +0.9 % over
    // the fixed stride: their decodes take tens of microseconds and a workgroup is expensive to start); the fixed stride on
    // the 16x grid for the smaller ones, where the hardware dispatcher is a queue that costs no atomics (the TC codes' draws
    // would hit the device's ceiling of ~85 M same-address atomics per second: claim_chunk()).
    const size_t resident = (size_t)resident_workgroups<CODE, T, IPT, PF, LEAN, FORM, NANPASS>();
    constexpr bool queue_fed = GEO::WG >= 512;
    uint32_t *claim = (static_stride || maxiters == 0 || !queue_fed) ? nullptr : claim_counter(stream);
    // groups per draw: at least ~8 draws per resident workgroup, so that the last chunks are a small part of a short launch
    size_t K = 1;
    if (claim != nullptr) {
        K = claim_chunk<CODE, T, IPT>();
        while (K > 1 && groups < 8 * K * resident) K /= 2;
    }
    const size_t chunks = (groups + K - 1) / K;
    size_t grid = (resident <= 256 || claim != nullptr) ? resident : resident * 16;
    if (grid > chunks) grid = chunks;
    if constexpr (NANPASS == 2) {                // one round of workgroups, each looking at 64 marks per load
        grid = resident < (groups + 63) / 64 ? resident : (groups + 63) / 64;
    }
    constexpr bool clamp_form = std::is_same_v<T, float> && FORM == 2;
    if constexpr (NANPASS == 0 && notify_kernel_built<CODE>()) {        // (a two-pass decode is two launches: never)
        uint32_t *notify = nullptr, notify_ticket = 0;
        if (take_notify(grid, notify, notify_ticket)) {
            hipLaunchKernelGGL((decode_ms_notify_kernel<CODE, T, IPT, PF, LEAN, FORM, NANPASS>), dim3(1), dim3(GEO::WG), 0, stream,
                               llrs, output, iters, success, (uint32_t)batch, maxiters, nocap_limit_for(maxiters, clamp_form), claim, (uint32_t)K,
                               notify, notify_ticket);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((decode_ms_kernel<CODE, T, IPT, PF, LEAN, FORM, NANPASS>), dim3((unsigned)grid), dim3(GEO::WG), 0, stream,
                       llrs, output, iters, success, (uint32_t)batch, maxiters, nocap_limit_for(maxiters, clamp_form), claim, (uint32_t)K);
    return hipGetLastError();
}

// IPT indices per thread; LEAN 1 = register-lean check phase, 2 = in-place messages.  The f32 kernels with a clamp-free loop
// exist in two forms of the self-correction: the v_fma clamp (faster: TM2048 40.3 against 38.2, TC512 140.0 against 136.6 M
// codewords/s) narrows the range vote to |LLR| <= 2^floor(82.5 - log2(7) max_iters) -- 2^12 at the benchmark's 25 iterations,
// 2^3 at 28; beyond that real LLRs would fall out of the clamp-free loop altogether, so longer decodes run the v_mul_legacy
// form, whose vote is the clamp-free loop's own.
//
// NaN LLRs in two passes.  The NaN -> +inf mapping of Ops<float>::load is free in every f32 kernel but one: the register-lean
// TM5120 kernel re-reads its LLRs in every variable phase and has no register to spare -- canonicalising the finished marginals
// instead costs it 45 spilled registers, 17.1 -> 14.4 M codewords/s.  Its batches therefore run two kernels on the stream: the
// first decodes as if no LLR were a NaN -- round 2's loop -- and looks at the marginals it has in registers when a decode ends
// (a NaN LLR leaves a NaN marginal, nothing else does): for a codeword with one it leaves NAN_MARK in `iters` instead of
// results.  The second is the NaN-handling kernel restricted to the marked codewords: without a NaN in the batch a few loads per
// workgroup.  TM5120 f32 14.4 -> 16.9 M codewords/s at 524 288 frames, +5 % still at 1 024; with a NaN in every hundredth frame
// 16.1 (profiles/r03_kbench/nan_two_pass_ab.txt).  Batches below two rounds of workgroups run the NaN-handling kernel alone, as do
// max_iters = 0 (nothing depends on the LLRs) and max_iters = 2^32 - 1 (the mark must not be an iteration count).  Results
// are identical either way (tests/test_gpu_parity.py::test_nan_two_pass).
// TM1280 f32 -- on the register-lean kernel since the same round -- does the same: its NaN-blind first pass also takes the peeled
// first iteration (which the NaN-handling form cannot afford): 72.9 -> 77.0 M codewords/s at 1 048 576 frames, +3 % still at
// 4 096.  The price is paid by batches that DO hold NaNs, whose marked codewords are a sparse second launch: with one in
// every hundredth frame 59.2 against 68.0 at 65 536 frames, 71.5 against 72.3 at 1 048 576 (TM5120: 16.1 against 14.3 -- still
// ahead).  NaN LLRs are a defect upstream of a decoder, not a workload; the common case gets the 5 %.
template <int CODE, class T, int IPT, int LEAN>
constexpr bool two_pass_nan()
{
    return std::is_same_v<T, float> && IPT == 1 && (CODE == TM5120 || CODE == TM1280) && LEAN == 1;
}

template <int CODE, class T, int IPT, int LEAN>
hipError_t launch_cfg(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                      size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    if constexpr (two_pass_nan<CODE, T, IPT, LEAN>()) {
        constexpr int FORM = selfcorr_med3<CODE, T>();
        static_assert(!has_nocap_loop<CODE, T, IPT, LEAN>());
        bool two = batch >= 2 * (size_t)resident_workgroups<CODE, T, IPT, false, LEAN, FORM, 1>();
        if (lflags & LF_TWO_PASS) two = true;
        if ((lflags & LF_ONE_PASS) || maxiters == 0 || maxiters == NAN_MARK) two = false;
        if (two) {
            const hipError_t e = launch_cfg_form<CODE, T, IPT, LEAN, FORM, 1>(llrs, output, iters, success, batch, maxiters, stream, lflags);
            if (e != hipSuccess) return e;
            return launch_cfg_form<CODE, T, IPT, LEAN, FORM, 2>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        }
        return launch_cfg_form<CODE, T, IPT, LEAN, FORM>(llrs, output, iters, success, batch, maxiters, stream, lflags);
    } else
    if constexpr (has_nocap_loop<CODE, T, IPT, LEAN>() && selfcorr_med3<CODE, T>() == 2) {
        if (nocap_limit_for(maxiters, true) >= 8.0f)
            return launch_cfg_form<CODE, T, IPT, LEAN, 2>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        return launch_cfg_form<CODE, T, IPT, LEAN, 3>(llrs, output, iters, success, batch, maxiters, stream, lflags);
    } else {
        return launch_cfg_form<CODE, T, IPT, LEAN, selfcorr_med3<CODE, T>()>(llrs, output, iters, success, batch, maxiters, stream, lflags);
    }
}

// Which form of the one-index-per-thread kernel a (code, type) runs: 0 = plain, 1 = register-lean check phase.
// Register-lean variant (decode_ms_kernel.hpp): pays where it doubles the workgroups per CU, which
// is TM5120 (f32 12.4 -> 13.8, i8 11.3 -> 13.8 M codewords/s at 4 dB) and the narrow types of TM1280.  Measured slower
// on TM1280 / TM1536 / TM2048 / TM6144 f32 (-7..-8 %), whose occupancy it does not change.
// TM1280 i8 / i16: 168 -> 116 VGPRs, four waves per SIMD instead of three: 68.4 -> 71.0 (its f32 kernel 70.5 -> 69.5: not).
// Re-measured on round 3's kernels (the lean check phase lost 70 VALU instructions per thread and iteration this round:
// profiles/r03_kbench/kb27_lean_again.txt), lean against plain, M codewords/s: TM1280 f32 69.97 -> 74.48 (115 instead of 168
// registers: 8 workgroups per CU instead of 6); TM2048 i8 51.47 -> 53.80 (64 instead of 80: 4 instead of 3); TM1536 i8
// 65.84 -> 67.88 (5 instead of 4); still slower where it does not buy occupancy: TM1536 f32 69.3 -> 57.7, TM2048 f32 40.2 ->
// 31.0, TM6144 f32 11.94 -> 9.68, i8 12.41 -> 11.79.
template <int CODE, class T, int IPT>
constexpr int lean_mode()
{
    constexpr bool narrow = std::is_same_v<T, int8_t> || std::is_same_v<T, int16_t>;
    constexpr bool lean_code = CODE == TM5120 || (CODE == TM1280 && (narrow || std::is_same_v<T, float>)) || ((CODE == TM1536 || CODE == TM2048) && narrow);
    return lean_code && IPT == 1 && !std::is_same_v<T, int32_t> ? 1 : 0;   // (i32's wider integer sequences spill at the lean kernel's 128-VGPR budget)
}

// Does a kernel form read every LLR from memory exactly once?  The plain kernels and the pair kernel hold their LLRs in
// registers; the register-lean kernel does so only for the narrow types (packed raw bytes), its f32 / f64 / i32 forms and the
// in-place form (LEAN 2) re-read them in every variable phase.  capi.hip lets the smallest host calls read their input
// across the link only when this holds (one predicate, next to the tables that pick the kernel: round 3 advice).
template <class T, int LEAN>
constexpr bool kernel_reads_llrs_once()
{
    return LEAN == 0 || (LEAN == 1 && (std::is_same_v<T, int8_t> || std::is_same_v<T, int16_t>));
}

template <int CODE, class T, int IPT>
hipError_t launch_one(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                      size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    return launch_cfg<CODE, T, IPT, lean_mode<CODE, T, IPT>()>(llrs, output, iters, success, batch, maxiters, stream, lflags);
}

// Pair-ownership kernel (decode_ms_pair.hpp): one workgroup per CU-resident codeword, persistent.
constexpr int VARIANT_PAIR = 32;
template <int CODE, class T, int FORM>
hipError_t launch_pair_form(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                            size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    using GEO = PairGeometry<CODE, T>;
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int resident = cached[dev].load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_pair_kernel<CODE, T, FORM>, GEO::NT, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = per_cu * cus;
        cached[dev].store(resident, std::memory_order_relaxed);
    }
    uint32_t *claim = ((lflags & LF_STATIC) != 0 || maxiters == 0) ? nullptr : claim_counter(stream);
    size_t grid = (resident <= 256 || claim != nullptr) ? (size_t)resident : (size_t)resident * 16;
    if (grid > batch) grid = batch;
    constexpr bool clamp_form = std::is_same_v<T, float> && FORM == 2;
    hipLaunchKernelGGL((decode_ms_pair_kernel<CODE, T, FORM>), dim3((unsigned)grid), dim3(GEO::NT), 0, stream,
                       llrs, output, iters, success, (uint32_t)batch, maxiters, nocap_limit_for(maxiters, clamp_form), claim);
    return hipGetLastError();
}

template <int CODE, class T>
hipError_t launch_pair(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                       size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    if (batch == 0) return hipSuccess;
    if (batch > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if constexpr (std::is_same_v<T, float>) {
        // f32: the v_fma clamp form (+10 %) narrows the range vote to |LLR| <= 2^floor(82.5 - log2(7) max_iters) -- 2^12 at the
        // benchmark's 25 iterations, 2^3 at 28; beyond that real LLRs would fall out of the clamp-free loop altogether, so longer
        // decodes run the v_mul_legacy form (+7 %), whose vote is the clamp-free loop's own
        if (nocap_limit_for(maxiters, true) >= 8.0f)
            return launch_pair_form<CODE, T, 2>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        return launch_pair_form<CODE, T, 3>(llrs, output, iters, success, batch, maxiters, stream, lflags);
    } else {
        return launch_pair_form<CODE, T, pair_form_default<T>()>(llrs, output, iters, success, batch, maxiters, stream, lflags);
    }
}

// one `case` of the dispatch switch: default IPT plus optional alternatives
// (expects `variant` with VARIANT_STATIC already split off into `static_stride`: LDPC_SPLIT_VARIANT)
#define LDPC_CASE(CODE, T, DEF, ...)                                                             \
    case CODE: {                                                                                 \
        constexpr int alts[] = {DEF, ##__VA_ARGS__};                                             \
        return dispatch_ipt<CODE, T, DEF, ##__VA_ARGS__>(variant == 0 ? alts[0] : variant, llrs, \
                                                         output, iters, success, batch, maxiters, \
                                                         stream, lflags);                        \
    }
// the same table row as a `case` of decode_ms_reads_llrs_once<T>(): the DEFAULT kernel of the code
#define LDPC_ONCE_CASE(CODE, T, DEF, ...)                                                        \
    case CODE: return kernel_reads_llrs_once<T, lean_mode<CODE, T, DEF>()>();
#define LDPC_SPLIT_VARIANT()                                                                     \
    unsigned lflags = 0;                                                                         \
    if (variant >= 0) {                                                                          \
        if (variant & VARIANT_STATIC) lflags |= LF_STATIC;                                       \
        if (variant & VARIANT_ONE_PASS) lflags |= LF_ONE_PASS;                                   \
        if (variant & VARIANT_TWO_PASS) lflags |= LF_TWO_PASS;                                   \
        variant &= ~VARIANT_FLAGS;                                                               \
    }

template <int CODE, class T, int... IPTS>
hipError_t dispatch_ipt(int ipt, const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                        size_t batch, uint32_t maxiters, hipStream_t stream, unsigned lflags)
{
    hipError_t r = hipErrorInvalidConfiguration;
    (void)((ipt == IPTS ? (r = launch_one<CODE, T, IPTS>(llrs, output, iters, success, batch, maxiters, stream, lflags), true)
                        : false) || ...);
    return r;
}

}  // namespace ldpc
