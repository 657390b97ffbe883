// channel.hip -- synthetic BPSK + AWGN frames, generated on the device.
//
// Harness side of the path: what perftest's ms_trial does per frame on the CPU
// (/root/reference/perftest/src/main.rs:10-18: encode, hard_to_llrs -> +-1, add Normal noise)
// done for a whole batch in HBM, so that the 16 GiB/GPU of LLRs of the TM8192 configuration
// never cross PCIe.  One thread makes four consecutive samples: one Philox4x32-10 block
// keyed by the seed and counted by (sample/4, GLOBAL frame index), two Box-Muller pairs, one 16-byte
// (f32) or 4-byte (i8) store -- fully coalesced, HBM-write bound.  The global frame index (first_frame +
// the frame's position in the call) also picks the codeword from the pool, so a call that generates frames
// [a, b) of a job writes exactly bytes [a, b) of the buffer a single call for the whole job would write:
// the shards of an N-GPU run are slices of the one-GPU batch, bit for bit (labrador_ldpc_hip_awgn_*_at).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <vector>

#include "channel.hpp"

namespace ldpc {

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
        k0 += W0;
        k1 += W1;
    }
    return ctr;
}

// two standard normals from two 32-bit words (Box-Muller; u1 in (0,1], u2 in [0,1))
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1)
{
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

template <class T> struct Quant;
template <> struct Quant<float> {
    static __device__ __forceinline__ float q(float y, float, int) { return y; }
};
template <> struct Quant<int8_t> {
    static __device__ __forceinline__ int8_t q(float y, float scale, int lim)
    {
        int v = (int)rintf(scale * y);
        v = v < -lim ? -lim : (v > lim ? lim : v);
        return (int8_t)v;
    }
};

template <class T>
__global__ void __launch_bounds__(256)
awgn_kernel(const uint8_t *__restrict__ codewords, uint32_t pool, T *__restrict__ llrs,
            uint32_t n, uint64_t first_frame, uint64_t total_quads, float sigma, float scale, int lim,
            uint32_t seed_lo, uint32_t seed_hi)
{
    const uint32_t quads_per_frame = n / 4;
    for (uint64_t qd = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < total_quads;
         qd += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t local = qd / quads_per_frame;
        const uint32_t q = (uint32_t)(qd - local * quads_per_frame);
        const uint64_t frame = first_frame + local;
        const U4 rnd = philox4x32_10(U4{q, (uint32_t)frame, (uint32_t)(frame >> 32), 0u}, seed_lo, seed_hi);
        float z[4];
        box_muller(rnd.x, rnd.y, z[0], z[1]);
        box_muller(rnd.z, rnd.w, z[2], z[3]);
        const uint32_t cwi = (uint32_t)(frame % pool);
        const uint32_t byte = codewords[(size_t)cwi * (n / 8) + q / 2];
        const uint32_t nib = (q & 1) ? (byte & 0xF) : (byte >> 4);      // MSB-first bits 4q..4q+3
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = ((nib >> (3 - j)) & 1) ? -1.0f : 1.0f;      // bit 1 -> -1 (decoder.rs:487-490)
            o[j] = Quant<T>::q(s + sigma * z[j], scale, lim);
        }
        T *dst = llrs + qd * 4;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(dst) = float4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        } else {
            *reinterpret_cast<char4 *>(dst) = char4{(char)o[0], (char)o[1], (char)o[2], (char)o[3]};
        }
    }
}

// Harness diagnostic: the shader clock the chip holds under a VALU load.  Every wave spins on a dependent chain for a fixed
// WALL time (s_memrealtime, 100 MHz) and reports how far s_memtime -- which counts shader clocks on gfx950
// (tools/ubench/clock_rate.hip) -- advanced meanwhile.  A bench line carries the figure per rank, so that a slow rank of an
// N-GPU run is explained by the line itself (a chip that clocks lower) rather than by a second run.
__global__ void __launch_bounds__(256) clock_probe_kernel(unsigned long long *out, unsigned long long wall_ticks)
{
    float a = threadIdx.x * 1.5f + 1.f, b = a * 3.f + 1.f, c = b - 7.f, d = a + b;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1;
    do {
        for (int l = 0; l < 64; ++l)
            asm volatile("v_min_f32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_min_f32 %2, %3, %2\n v_xor_b32 %3, %0, %3\n"
                         "v_add_f32 %0, %1, %0\n v_mul_f32 %1, %2, %1\n v_min3_f32 %2, %3, %2, %0\n v_sub_f32 %3, %0, %3"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < wall_ticks);
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = m1 - m0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (a + b + c + d == 12345.f) out[2 * gridDim.x] = 1;      // keeps the chain alive
}

}  // namespace

hipError_t shader_clock_mhz(double busy_ms, double *mhz)
{
    constexpr unsigned BLOCKS = 256 * 8;                       // two 256-thread workgroups per SIMD quartet: 8 waves per SIMD
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc(&d, (2 * BLOCKS + 1) * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    if (busy_ms < 0.01) busy_ms = 0.01;
    if (busy_ms > 1000.0) busy_ms = 1000.0;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(BLOCKS), dim3(256), 0, nullptr, d, (unsigned long long)(busy_ms * 1e5));
    e = hipGetLastError();
    std::vector<unsigned long long> h(2 * BLOCKS);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    std::vector<double> r;
    for (unsigned b = 0; b < BLOCKS; ++b)
        if (h[2 * b + 1]) r.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
    if (r.empty()) return hipErrorUnknown;
    std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
    *mhz = r[r.size() / 2];                                    // median over the workgroups
    return hipSuccess;
}

template <class T>
hipError_t launch_awgn(const uint8_t *codewords, size_t pool, T *llrs, int n, uint64_t first_frame, size_t batch, float sigma,
                       float scale, int lim, uint64_t seed, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const uint64_t total_quads = (uint64_t)batch * (n / 4);
    uint64_t blocks = (total_quads + 255) / 256;
    if (blocks > 256ull * 32) blocks = 256ull * 32;
    hipLaunchKernelGGL((awgn_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, codewords,
                       (uint32_t)pool, llrs, (uint32_t)n, first_frame, total_quads, sigma, scale, lim,
                       (uint32_t)seed, (uint32_t)(seed >> 32));
    return hipGetLastError();
}

template hipError_t launch_awgn<float>(const uint8_t *, size_t, float *, int, uint64_t, size_t, float, float, int, uint64_t, hipStream_t);
template hipError_t launch_awgn<int8_t>(const uint8_t *, size_t, int8_t *, int, uint64_t, size_t, float, float, int, uint64_t, hipStream_t);

}  // namespace ldpc
