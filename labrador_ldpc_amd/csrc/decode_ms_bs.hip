// decode_ms_bs.hip -- gfx950 instantiation of the bit-sliced i8 min-sum decoder (decode_ms_bitslice.hpp;
// decode_ms::<i8>, /root/reference/src/decoder.rs:42-50, :347-475) for the TM codes, and its launcher.
//
// One wave per workgroup, one group of G = 64 / (M/32) codewords per wave at a time: a codeword never leaves its wave, so the
// kernel has no barrier.  Rate 1/2 and 2/3: one group per workgroup -- the hardware dispatcher is the work queue (decodes take
// 3..25 iterations; a finished wave makes room for the next group at once); LDS per workgroup: the LLRs as bit planes (n bytes per
// codeword, the size of the raw LLRs), a 2 KB staging slab, the lane-permutation table, one word per lane and block column for
// the hard decisions.  Rate 4/5: persistent waves (the resident set), each with a 20 KB slot of a stream-ordered global
// workspace for its LLR planes, groups drawn from a queue head at the start of that workspace (decode_ms_bitslice.hpp, "register diet").
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

#include "decode_ms_bitslice.hpp"
#include "decode_ms_bitslice_split.hpp"
#include "hip_backend.hpp"

#if defined(BS_DIAG) && !defined(BS_DIAG_BUILD)
#error "BS_DIAG builds decode wrongly: only tools/bs_diag_build.sh may define it"
#endif

namespace ldpc {
namespace bs {


// waves per SIMD the kernel is compiled for: two for every code (256 registers).  The rate-1/2 and rate-2/3 codes hold a group of
// codewords in that; the rate-4/5 codes' state (39 edges: ~250 planes before any temporary) is put on the "register diet" of
// decode_ms_bitslice.hpp to fit -- one wave per SIMD with everything in registers measured 23.4 against 33.3 M codewords/s (TM5120)
template <int CODE> constexpr int waves_per_simd() { return 2; }

constexpr int WS_HEADER_WORDS = 64;          // the queue head, on a 256-byte line of its own

template <int CODE>
__global__ void __launch_bounds__(64, waves_per_simd<CODE>())
decode_ms_bs_kernel(const int8_t *__restrict__ llrs, uint8_t *__restrict__ output, uint32_t *__restrict__ iters,
                    uint8_t *__restrict__ success, uint32_t batch, uint32_t maxiters, uint32_t ngroups, uint32_t *__restrict__ workspace)
{
    __shared__ __attribute__((aligned(16))) char lds[Geo<CODE>::LDS_BYTES];
    HipBackend b{lds};
    init_kernel<CODE, HipBackend>(b);
    if constexpr (Geo<CODE>::LLR_GLOBAL) {
        // persistent waves, fed from a queue: the workspace starts with the queue head (zeroed by the launcher), this wave's slot of LLR
        // planes follows.  A wave's first group is its own index; every further one is drawn with one atomic.  (A fixed stride lost 7-14 %
        // even where every frame takes the same 25 iterations -- the waves of a CU do not run at one speed -- and 40 % on a box whose
        // chip was unevenly clocked: profiles/r04_kbench/r45_queue_ab.txt.)
        uint32_t *ws = workspace + WS_HEADER_WORDS + (size_t)blockIdx.x * Geo<CODE>::LLR_WORDS;
        uint32_t g = blockIdx.x;
        while (g < ngroups) {
            decode_group<CODE, HipBackend>(b, llrs, output, iters, success, batch, maxiters, g, ws);
            uint32_t t = 0;
            if ((threadIdx.x & 63) == 0) t = atomicAdd(workspace, 1u);
            g = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        }
    } else {
        for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) decode_group<CODE, HipBackend>(b, llrs, output, iters, success, batch, maxiters, g, nullptr);
    }
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

template <int CODE>
hipError_t launch(const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const size_t groups = (batch + G - 1) / G;
    size_t grid = groups < 0x7FFFFFFFull ? groups : 0x7FFFFFFFull;
    uint32_t *ws = nullptr;
    if constexpr (Geo<CODE>::LLR_GLOBAL) {
        // persistent waves, one workspace slot each: the grid is the resident set, the groups are drawn from a queue (kernel)
        static std::atomic<int> cached[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        int resident = cached[dev].load(std::memory_order_relaxed);
        if (resident == 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_bs_kernel<CODE>, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            resident = per_cu * cus;
            cached[dev].store(resident, std::memory_order_relaxed);
        }
        if (grid > (size_t)resident) grid = (size_t)resident;
        hipError_t e = hipMallocAsync((void **)&ws, (WS_HEADER_WORDS + grid * Geo<CODE>::LLR_WORDS) * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(ws, 0, WS_HEADER_WORDS * sizeof(uint32_t), stream);
        if (e != hipSuccess) { (void)hipFreeAsync(ws, stream); return e; }
    }
    hipLaunchKernelGGL((decode_ms_bs_kernel<CODE>), dim3((unsigned)grid), dim3(64), 0, stream, llrs, output, iters, success, (uint32_t)batch, maxiters,
                       (uint32_t)groups, ws);
    hipError_t e = hipGetLastError();
    if (ws != nullptr) {
        const hipError_t e2 = hipFreeAsync(ws, stream);
        if (e == hipSuccess) e = e2;
    }
    return e;
}

}  // namespace bs

// the rate-4/5 codes through the two-waves-per-group kernel; hipErrorInvalidConfiguration for every other code
hipError_t launch_decode_ms_bitsliced_split(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                            uint32_t maxiters, hipStream_t stream)
{
    switch (code) {
        case TM1280: return bs::launch_split<TM1280>(llrs, output, iters, success, batch, maxiters, stream);
        case TM5120: return bs::launch_split<TM5120>(llrs, output, iters, success, batch, maxiters, stream);
        default: return hipErrorInvalidConfiguration;
    }
}

// i8 LLRs through the bit-sliced kernel; hipErrorInvalidConfiguration for the codes it is not built for (the TC codes: their
// circulants are not quarter-wise rotations).  llrs 4-byte aligned, output 4-byte aligned.
hipError_t launch_decode_ms_bitsliced(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream)
{
    switch (code) {
        case TM1280: return bs::launch<TM1280>(llrs, output, iters, success, batch, maxiters, stream);
        case TM1536: return bs::launch<TM1536>(llrs, output, iters, success, batch, maxiters, stream);
        case TM2048: return bs::launch<TM2048>(llrs, output, iters, success, batch, maxiters, stream);
        case TM5120: return bs::launch<TM5120>(llrs, output, iters, success, batch, maxiters, stream);
        case TM6144: return bs::launch<TM6144>(llrs, output, iters, success, batch, maxiters, stream);
        case TM8192: return bs::launch<TM8192>(llrs, output, iters, success, batch, maxiters, stream);
        default: return hipErrorInvalidConfiguration;
    }
}

}  // namespace ldpc
