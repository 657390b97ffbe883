// decode_ms_bs.hip -- gfx950 instantiation of the bit-sliced i8 min-sum decoder (decode_ms_bitslice.hpp;
// decode_ms::<i8>, /root/reference/src/decoder.rs:42-50, :347-475) for the TM codes, and its launcher.
//
// Rate 1/2 and 2/3: one wave per workgroup, one group of G = 64 / (M/32) codewords per wave at a time -- a codeword never leaves its
// wave, so the kernel has no barrier; one group per workgroup: the hardware dispatcher is the work queue (decodes take 3..25
// iterations; a finished wave makes room for the next group at once).  LDS per workgroup: the LLRs as bit planes (n bytes per
// codeword, the size of the raw LLRs), the lane-permutation table, one word per lane and block column for the hard decisions (which
// doubles as the staging slab of the LLR transposition).  Rate 4/5: a group shared by the two waves of a workgroup
// (decode_ms_bitslice_split.hpp).
//
// Compiled TWICE (Makefile, -DBS_TU): unit 1 = the rate-1/2 kernels and the dispatcher, with the default machine scheduler; unit 2 =
// the rate-2/3 kernels and the two-wave kernels of the rate-4/5 codes, with -mllvm -amdgpu-sched-strategy=iterative-ilp, which is
// worth +2 % on the two-wave kernels and +0.5 % on rate 2/3 but costs the rate-1/2 kernels (168 registers, spilling) 2.5 %
// (profiles/r05_kbench/permute_pipeline.txt, section 7).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

#ifndef BS_TU
#error "decode_ms_bs.hip is compiled per unit: -DBS_TU=1 (rate 1/2 + dispatch) or -DBS_TU=2 (rate 2/3, rate 4/5)"
#endif

#include "decode_ms_bitslice.hpp"
#include "decode_ms_bitslice_split.hpp"
#include "hip_backend.hpp"

namespace ldpc {
namespace bs {

// waves per SIMD the kernel is compiled for.  Rate 1/2 (TM2048, TM8192): THREE -- 168 registers -- since round 5: a row's running
// state replaces its old state with the row's last edge, block row 0 finishes first, the hard decisions live in LDS
// (decode_ms_bitslice.hpp, "schedule of an iteration").  Rate 2/3: two (256 registers).
#ifndef BS_WAVES_R12
#define BS_WAVES_R12 3
#endif
template <int CODE> constexpr int waves_per_simd() { return (CODE == TM2048 || CODE == TM8192) ? BS_WAVES_R12 : 2; }

template <int CODE>
__global__ void __launch_bounds__(64, waves_per_simd<CODE>())
decode_ms_bs_kernel(const int8_t *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters,
                    uint8_t *__restrict__ success, uint32_t batch, uint32_t maxiters, uint32_t ngroups)
{
    __shared__ __attribute__((aligned(16))) char lds[Geo<CODE>::LDS_BYTES];
    HipBackend b{lds};
    init_kernel<CODE, HipBackend>(b);
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) decode_group<CODE, HipBackend>(b, llrs, output, iters, success, batch, maxiters, g);
}

// ---- slot refill (decode_refill, decode_ms_bitslice.hpp): persistent waves, frames handed out in chunks through the launch's queue word
// (one relaxed atomic per CHUNK frames; the word is zero between launches: every wave draws until its first ticket beyond the last chunk,
// and the holder of the very last ticket puts the word back).  A finished slot takes the wave's next frame at once.
template <int CODE>
__global__ void __launch_bounds__(64, waves_per_simd<CODE>())
decode_ms_bs_refill_kernel(const int8_t *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters,
                           uint8_t *__restrict__ success, uint32_t batch, uint32_t maxiters, uint32_t *queue, uint32_t chunk)
{
    __shared__ __attribute__((aligned(16))) char lds[Geo<CODE>::LDS_BYTES];
    HipBackend b{lds};
    init_kernel<CODE, HipBackend>(b);
    const uint32_t nchunks = (batch + chunk - 1) / chunk;
    uint32_t cur = 0, end = 0;
    auto next = [&]() -> uint32_t {
        if (cur == end) {
            uint32_t t = 0;
            if (threadIdx.x == 0) t = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= nchunks) {                                                     // (decode_refill asks no more after this)
                if (t == nchunks + gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return NO_FRAME;
            }
            cur = t * chunk;
            end = batch - cur < chunk ? batch : cur + chunk;
        }
        return cur++;
    };
    decode_refill<CODE, HipBackend>(b, llrs, output, iters, success, maxiters, next);
}

// ---- the rate-4/5 codes: a group of codewords shared by the two waves of a workgroup (decode_ms_bitslice_split.hpp).  One group per
// workgroup: the hardware dispatcher is the work queue, as for the rate-1/2 and rate-2/3 kernel above. ----
template <int CODE, int HALF>
__device__ __forceinline__ void split_wave(char *lds, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, uint32_t batch,
                                           uint32_t maxiters, uint32_t ngroups)
{
    HipBackend b{lds + SplitLayout<CODE>::template priv<HALF>()};
    SplitGroup<CODE, HipBackend, HALF> g;
    g.init(b);
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x)
        decode_group_split<CODE, HipBackend, HALF>(b, g, llrs, output, iters, success, batch, maxiters, grp, [] { __syncthreads(); });
}

template <int CODE>
__global__ void __launch_bounds__(128, 2)
decode_ms_bs_split_kernel(const int8_t *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters, uint8_t *__restrict__ success,
                          uint32_t batch, uint32_t maxiters, uint32_t ngroups)
{
    __shared__ __attribute__((aligned(16))) char lds[SplitLayout<CODE>::BYTES];
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) split_wave<CODE, 0>(lds, llrs, output, iters, success, batch, maxiters, ngroups);
    else split_wave<CODE, 1>(lds, llrs, output, iters, success, batch, maxiters, ngroups);
}

// ---- slot refill on the two-wave kernel (decode_refill_split): persistent workgroups; wave 0 draws chunks of frames from the launch's
// queue word and hands the ticket to wave 1 through an LDS word behind a barrier (both waves ask at the same points of their identical
// bookkeeping); the word is self-resetting like decode_ms_bs_refill_kernel's: every workgroup draws until its first ticket beyond the last chunk.
template <int CODE, int HALF>
__device__ __forceinline__ void split_wave_refill(char *lds, uint32_t *ticket_word, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                                                  uint32_t batch, uint32_t maxiters, uint32_t *queue, uint32_t chunk)
{
    HipBackend b{lds + SplitLayout<CODE>::template priv<HALF>()};
    SplitGroup<CODE, HipBackend, HALF> g;
    g.init(b);
    const uint32_t nchunks = (batch + chunk - 1) / chunk;
    uint32_t draws = 0;
    auto draw = [&]() -> uint64_t {
        // two ticket words, used alternately: wave 0 may write draw k + 1's ticket before wave 1 has read draw k's (only the barrier of
        // draw k + 1 lies between them), never draw k + 2's
        uint32_t *word = ticket_word + (draws++ & 1u);
        if (HALF == 0 && threadIdx.x == 0) {
            const uint32_t t = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == nchunks + gridDim.x - 1) __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the launch's last draw
            *word = t;
        }
        __syncthreads();
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)*word);
        if (t >= nchunks) return 0;
        const uint32_t lo = t * chunk, hi = batch - lo < chunk ? batch : lo + chunk;
        return (uint64_t)lo | (uint64_t)hi << 32;
    };
    decode_refill_split<CODE, HipBackend, HALF>(b, g, llrs, output, iters, success, maxiters, [] { __syncthreads(); }, draw);
}

template <int CODE>
__global__ void __launch_bounds__(128, 2)
decode_ms_bs_split_refill_kernel(const int8_t *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters, uint8_t *__restrict__ success,
                                 uint32_t batch, uint32_t maxiters, uint32_t *queue, uint32_t chunk)
{
    __shared__ __attribute__((aligned(16))) char lds[SplitLayout<CODE>::BYTES];
    __shared__ uint32_t ticket_word[2];
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) split_wave_refill<CODE, 0>(lds, ticket_word, llrs, output, iters, success, batch, maxiters, queue, chunk);
    else split_wave_refill<CODE, 1>(lds, ticket_word, llrs, output, iters, success, batch, maxiters, queue, chunk);
}

template <int CODE>
hipError_t launch_split_refill(const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream,
                               uint32_t *queue)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int resident = cached[dev].load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_bs_split_refill_kernel<CODE>, 128, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = per_cu * cus;
        cached[dev].store(resident, std::memory_order_relaxed);
    }
    // frames per draw: four rounds of slots, fewer for short launches (at least ~4 draws per resident workgroup)
    size_t chunk = 4 * (size_t)G;
    while (chunk > (size_t)G && batch / chunk < 4 * (size_t)resident) chunk /= 2;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    const size_t grid = nchunks < (size_t)resident ? nchunks : (size_t)resident;
    hipLaunchKernelGGL((decode_ms_bs_split_refill_kernel<CODE>), dim3((unsigned)grid), dim3(128), 0, stream, llrs, output, iters, success, (uint32_t)batch, maxiters,
                       queue, (uint32_t)chunk);
    return hipGetLastError();
}

template <int CODE>
hipError_t launch_split(const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const size_t groups = (batch + G - 1) / G;
    const size_t grid = groups < 0x7FFFFFFFull ? groups : 0x7FFFFFFFull;
    hipLaunchKernelGGL((decode_ms_bs_split_kernel<CODE>), dim3((unsigned)grid), dim3(128), 0, stream, llrs, output, iters, success, (uint32_t)batch, maxiters,
                       (uint32_t)groups);
    return hipGetLastError();
}

uint32_t *bs_queue_word(hipStream_t stream);          // decode_ms_i8.hip: the launch queue's word of (device, stream), or nullptr (claim_counter)

template <int CODE>
hipError_t launch_refill(const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream,
                         uint32_t *queue)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int resident = cached[dev].load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_bs_refill_kernel<CODE>, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = per_cu * cus;
        cached[dev].store(resident, std::memory_order_relaxed);
    }
    // frames per draw: 8 rounds of slots, fewer for short launches (at least ~4 draws per resident wave)
    size_t chunk = 8 * (size_t)G;
    while (chunk > (size_t)G && batch / chunk < 4 * (size_t)resident) chunk /= 2;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    const size_t grid = nchunks < (size_t)resident ? nchunks : (size_t)resident;
    hipLaunchKernelGGL((decode_ms_bs_refill_kernel<CODE>), dim3((unsigned)grid), dim3(64), 0, stream, llrs, output, iters, success, (uint32_t)batch, maxiters,
                       queue, (uint32_t)chunk);
    return hipGetLastError();
}

template <int CODE>
hipError_t launch(const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const size_t groups = (batch + G - 1) / G;
    const size_t grid = groups < 0x7FFFFFFFull ? groups : 0x7FFFFFFFull;
    hipLaunchKernelGGL((decode_ms_bs_kernel<CODE>), dim3((unsigned)grid), dim3(64), 0, stream, llrs, output, iters, success, (uint32_t)batch, maxiters,
                       (uint32_t)groups);
    return hipGetLastError();
}

}  // namespace bs

// i8 LLRs through the bit-sliced kernel; hipErrorInvalidConfiguration for the codes it is not built for (the TC codes: their
// circulants are not quarter-wise rotations).  llrs 4-byte aligned, output 4-byte aligned (the caller checks).  Rate 4/5: two waves per group.
// refill: 1 = slot refill where the code has it (bitslice_refills()), 0 = the lockstep kernels
hipError_t launch_decode_ms_bitsliced_unit2(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                            uint32_t maxiters, hipStream_t stream, int refill);
#if BS_TU == 2
hipError_t launch_decode_ms_bitsliced_unit2(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                            uint32_t maxiters, hipStream_t stream, int refill)
{
    uint32_t *queue = refill ? bs::bs_queue_word(stream) : nullptr;
    switch (code) {
        case TM1280:
            if (queue) return bs::launch_split_refill<TM1280>(llrs, output, iters, success, batch, maxiters, stream, queue);
            return bs::launch_split<TM1280>(llrs, output, iters, success, batch, maxiters, stream);
        case TM1536:
            if (queue) return bs::launch_refill<TM1536>(llrs, output, iters, success, batch, maxiters, stream, queue);
            return bs::launch<TM1536>(llrs, output, iters, success, batch, maxiters, stream);
        case TM5120:
#ifndef BS_REFILL_TM5120
#define BS_REFILL_TM5120 0
#endif
#if BS_REFILL_TM5120
            if (queue) return bs::launch_split_refill<TM5120>(llrs, output, iters, success, batch, maxiters, stream, queue);
#endif
            return bs::launch_split<TM5120>(llrs, output, iters, success, batch, maxiters, stream);
        case TM6144: return bs::launch<TM6144>(llrs, output, iters, success, batch, maxiters, stream);
        default: return hipErrorInvalidConfiguration;
    }
}
#else
hipError_t launch_decode_ms_bitsliced(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream, int refill)
{
    // (slot refill on TM2048, G = 4: built, bit-exact, -1 ... -4 %: its codeword passes the staging slab in two pieces -- the "v != 0"
    // planes of the active slots live behind the hard-decision words -- and four slots lose little to the lockstep:
    // profiles/r06_kbench/slot_refill_r06.txt.  -DBS_REFILL_R12=1 builds it.)
#ifndef BS_REFILL_R12
#define BS_REFILL_R12 0
#endif
    switch (code) {
        case TM2048:
#if BS_PLANES == 8 && BS_REFILL_R12
            if (uint32_t *queue = refill ? bs::bs_queue_word(stream) : nullptr)
                return bs::launch_refill<TM2048>(llrs, output, iters, success, batch, maxiters, stream, queue);
#endif
            return bs::launch<TM2048>(llrs, output, iters, success, batch, maxiters, stream);
        case TM8192: return bs::launch<TM8192>(llrs, output, iters, success, batch, maxiters, stream);
#if BS_PLANES == 8
        default: return launch_decode_ms_bitsliced_unit2(code, llrs, output, iters, success, batch, maxiters, stream, refill);
#else                   // (a 16-plane experiment build, tools/bs_alt_build.sh -DBS_PLANES=16, carries the rate-1/2 codes only)
        default: return hipErrorInvalidConfiguration;
#endif
    }
}
#endif

}  // namespace ldpc
