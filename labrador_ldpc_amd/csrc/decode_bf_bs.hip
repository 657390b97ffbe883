// decode_bf_bs.hip -- gfx950 instantiation of the bit-sliced hard-decision decoder (decode_bf_bitslice.hpp: LDPCCode::decode_bf,
// /root/reference/src/decoder.rs:243-301, with decode_erasures, :144-223) for the TM codes: one wave per group of 64 / (M/32) codewords.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "decode_bf_bitslice.hpp"
#include "hip_backend.hpp"

namespace ldpc {
namespace bs {

// groups per queue draw: four, except for the rate-4/5 codes -- their longer decodes (39 edges) stay far below the atomics' ceiling with
// one, and fewer, larger chunks cost them 12-15 % in the tail (TM5120 123 against 109 M codewords/s, TM1280 211 against 181)
template <int CODE> constexpr uint32_t bf_chunk() { return (CODE == TM1280 || CODE == TM5120) ? 1u : 4u; }

// waves per SIMD: four (128 registers), two for the rate-4/5 codes, whose 39 edges' permutations and 11 columns of counters do not fit 128
template <int CODE> constexpr int bf_waves_per_simd() { return (CODE == TM1280 || CODE == TM5120) ? 2 : 4; }

template <int CODE>
__global__ void __launch_bounds__(64, bf_waves_per_simd<CODE>())
decode_bf_bs_kernel(const uint8_t *__restrict__ input, uint8_t *__restrict__ output, uint32_t *__restrict__ iters, uint8_t *__restrict__ success,
                    uint32_t batch, uint32_t maxiters, uint32_t ngroups, uint32_t *__restrict__ queue)
{
    __shared__ __attribute__((aligned(16))) char lds[BfGeo<CODE>::LDS_BYTES];
    HipBackend b{lds};
    bf_init_kernel<CODE, HipBackend>(b);
    // persistent waves fed from a queue (a decode is a few microseconds and data dependent; the waves of a CU do not run at one speed).
    // A draw takes bf_chunk<CODE>() consecutive groups: same-address atomics complete at ~85 M per second on this device, which
    // one group per draw reaches -- TM2048 342 M codewords/s = 85 M groups/s, TM8192 85 M -- and four per draw stay clear of.
    constexpr uint32_t BF_CHUNK = bf_chunk<CODE>();
    const uint32_t nchunks = (ngroups + BF_CHUNK - 1) / BF_CHUNK;
    uint32_t c = blockIdx.x;
    while (c < nchunks) {
        const uint32_t g1 = (c + 1) * BF_CHUNK < ngroups ? (c + 1) * BF_CHUNK : ngroups;
        for (uint32_t g = c * BF_CHUNK; g < g1; ++g) bf_decode_group<CODE, HipBackend>(b, input, output, iters, success, batch, maxiters, g);
        uint32_t t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, 1u);
        c = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    }
}

template <int CODE>
hipError_t launch_bf(const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
    constexpr int G = Geo<CODE>::G;
    if (batch == 0) return hipSuccess;
    if (batch > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const size_t groups = (batch + G - 1) / G;
    // persistent waves: 16 per CU x 256 CUs cover the chip; the queue head is a stream-ordered 256-byte allocation
    const size_t chunks = (groups + bf_chunk<CODE>() - 1) / bf_chunk<CODE>();
    const size_t grid = chunks < 16384 ? chunks : 16384;
    uint32_t *queue = nullptr;
    hipError_t e = hipMallocAsync((void **)&queue, 256, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(queue, 0, 256, stream);
    if (e != hipSuccess) { (void)hipFreeAsync(queue, stream); return e; }
    hipLaunchKernelGGL((decode_bf_bs_kernel<CODE>), dim3((unsigned)grid), dim3(64), 0, stream, input, output, iters, success, (uint32_t)batch, maxiters,
                       (uint32_t)groups, queue);
    e = hipGetLastError();
    const hipError_t e2 = hipFreeAsync(queue, stream);
    return e != hipSuccess ? e : e2;
}

}  // namespace bs

// hipErrorInvalidConfiguration for the codes it is not built for (the TC codes).  input and output 4-byte aligned.
hipError_t launch_decode_bf_bitsliced(int code, const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream)
{
    switch (code) {
        case TM1280: return bs::launch_bf<TM1280>(input, output, iters, success, batch, maxiters, stream);
        case TM1536: return bs::launch_bf<TM1536>(input, output, iters, success, batch, maxiters, stream);
        case TM2048: return bs::launch_bf<TM2048>(input, output, iters, success, batch, maxiters, stream);
        case TM5120: return bs::launch_bf<TM5120>(input, output, iters, success, batch, maxiters, stream);
        case TM6144: return bs::launch_bf<TM6144>(input, output, iters, success, batch, maxiters, stream);
        case TM8192: return bs::launch_bf<TM8192>(input, output, iters, success, batch, maxiters, stream);
        default: return hipErrorInvalidConfiguration;
    }
}

}  // namespace ldpc
