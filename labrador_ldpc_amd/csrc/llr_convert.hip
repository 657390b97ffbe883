// llr_convert.hip -- the data formats either side of decode_ms, batched and device-resident:
//   hard_to_llrs (/root/reference/src/decoder.rs:484-493; C entries capi/src/lib.rs:129-153): packed hard
//   decisions, MSB first, to +-1 LLRs of type T, and llrs_to_hard (decoder.rs:498-509; capi/src/lib.rs:155-179).
// Both are flat maps over the batch (frames are whole numbers of bytes and lie back to back) and pure streaming:
// hard_to_llrs is bound by the HBM write of 8 * sizeof(T) bytes per byte read, llrs_to_hard by reading them.
// Every thread moves ONE 16-byte piece of the LLR array (K = 16 / sizeof(T) LLRs), so that a wave's accesses are
// 1 KB contiguous per instruction; the K-bit fields of the 8 / K lanes that share a byte are merged with DPP
// quad permutes.  Grid-stride loops, unrolled so that each thread has four pieces in flight.
#include "llr_convert.hpp"
#include <type_traits>

namespace ldpc {
namespace {

template <class T> struct alignas(16) Piece { T v[16 / sizeof(T)]; };

#ifndef LLRC_UNROLL
#define LLRC_UNROLL 4
#endif
#ifndef LLRC_NT_STORE
#define LLRC_NT_STORE 1
#endif
#ifndef LLRC_NT_LOAD
#define LLRC_NT_LOAD 1
#endif
constexpr int UNROLL = LLRC_UNROLL;

// int8_t: sixteen LLRs per piece cost ~64 VALU operations when built bit by bit; a 256-entry table in LDS (byte of
// bits -> its eight LLR bytes, 2 KB, built by the workgroup's 256 threads) makes a piece two 8-byte LDS reads:
// 4.7 -> 5.7 TB/s.
#ifndef LLRC_I8_TABLE
#define LLRC_I8_TABLE 1
#endif
__global__ void __launch_bounds__(256) hard_to_llrs_i8_table_kernel(const uint8_t *__restrict__ bits, int8_t *__restrict__ llrs, size_t pieces)
{
    __shared__ unsigned long long table[256];
    {
        unsigned long long e = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) e |= (unsigned long long)(((threadIdx.x >> (7 - j)) & 1u) ? 0xFFu : 0x01u) << (8 * j);   // decoder.rs:489-491
        table[threadIdx.x] = e;
    }
    __syncthreads();
    typedef int int4_ __attribute__((ext_vector_type(4)));
    const size_t i0 = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x;
    unsigned short two[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = i0 + u * 256;
        two[u] = i < pieces ? *reinterpret_cast<const unsigned short *>(bits + 2 * i) : (unsigned short)0;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = i0 + u * 256;
        if (i < pieces) {
            const unsigned long long lo = table[two[u] & 0xFFu], hi = table[two[u] >> 8];          // little-endian: the first byte is the low one
            const int4_ v = {(int)lo, (int)(lo >> 32), (int)hi, (int)(hi >> 32)};
            __builtin_nontemporal_store(v, reinterpret_cast<int4_ *>(llrs + i * 16));
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256) hard_to_llrs_kernel(const uint8_t *__restrict__ bits, T *__restrict__ llrs, size_t pieces)
{
    constexpr int K = 16 / sizeof(T);
    constexpr size_t stride = 256;                     // a workgroup converts UNROLL consecutive 4 KB spans of LLRs
    {
        const size_t i0 = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x;
        unsigned field[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
            field[u] = 0;
            if (i < pieces) {
                if constexpr (K == 16) field[u] = ((unsigned)bits[2 * i] << 8) | bits[2 * i + 1];
                else field[u] = ((unsigned)bits[(i * K) >> 3] >> (8 - K - (unsigned)((i * K) & 7))) & ((1u << K) - 1u);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
            if (i < pieces) {
                Piece<T> p;
#pragma unroll
                for (int j = 0; j < K; ++j) p.v[j] = ((field[u] >> (K - 1 - j)) & 1u) ? (T)-1 : (T)1;      // decoder.rs:489-491
#if LLRC_NT_STORE
                __attribute__((ext_vector_type(4))) int raw;
                __builtin_memcpy(&raw, &p, 16);
                __builtin_nontemporal_store(raw, reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(llrs + i * K));
#else
                *reinterpret_cast<Piece<T> *>(llrs + i * K) = p;
#endif
            }
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256) llrs_to_hard_kernel(const T *__restrict__ llrs, uint8_t *__restrict__ bits, size_t pieces)
{
    constexpr int K = 16 / sizeof(T);
    typedef int int4_ __attribute__((ext_vector_type(4)));
    constexpr size_t stride = 256;
    // `pieces` is a multiple of 8 / K and so is every lane's index modulo it: the lanes that share a byte are
    // active together
    {
        const size_t i0 = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x;
        int4_ raw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
#if LLRC_NT_LOAD
            if (i < pieces) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const int4_ *>(llrs + i * K));
#else
            if (i < pieces) raw[u] = *reinterpret_cast<const int4_ *>(llrs + i * K);
#endif
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
            if (i < pieces) {
                Piece<T> p;
                __builtin_memcpy(&p, &raw[u], 16);
                unsigned f = 0;
#pragma unroll
                for (int j = 0; j < K; ++j) f |= (p.v[j] < (T)0 ? 1u : 0u) << (K - 1 - j);                 // decoder.rs:504-506
                if constexpr (K == 16) {
                    *reinterpret_cast<uint16_t *>(bits + 2 * i) = (uint16_t)((f >> 8) | ((f & 0xFFu) << 8));
                } else if constexpr (K == 8) {
                    bits[i] = (uint8_t)f;
                } else {
                    // neighbour's field: quad_perm [1,0,3,2]; then the neighbouring pair's: quad_perm [2,3,0,1]
                    unsigned g = (f << K) | (unsigned)__builtin_amdgcn_update_dpp(0, (int)f, 0xB1, 0xF, 0xF, true);
                    if constexpr (K == 2) g = (g << 4) | (unsigned)__builtin_amdgcn_update_dpp(0, (int)g, 0x4E, 0xF, 0xF, true);
                    if ((i & (8 / K - 1)) == 0) bits[(i * K) >> 3] = (uint8_t)g;
                }
            }
        }
    }
}

inline size_t grid_for(size_t pieces) { return (pieces + 256 * UNROLL - 1) / (256 * UNROLL); }

}  // namespace

template <class T>
hipError_t launch_hard_to_llrs(const uint8_t *bits, T *llrs, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    constexpr size_t K = 16 / sizeof(T), SLICE = (size_t)1 << 38;             // pieces per launch: 2^30 workgroups of 256
    const size_t pieces = bytes * 8 / K;
    for (size_t p0 = 0; p0 < pieces; p0 += SLICE) {
        const size_t np = pieces - p0 < SLICE ? pieces - p0 : SLICE;
        if constexpr (std::is_same_v<T, int8_t> && LLRC_I8_TABLE)
            hipLaunchKernelGGL(hard_to_llrs_i8_table_kernel, dim3((unsigned)grid_for(np)), dim3(256), 0, stream, bits + p0 * K / 8, llrs + p0 * K, np);
        else
            hipLaunchKernelGGL(hard_to_llrs_kernel<T>, dim3((unsigned)grid_for(np)), dim3(256), 0, stream, bits + p0 * K / 8, llrs + p0 * K, np);
    }
    return hipGetLastError();
}

template <class T>
hipError_t launch_llrs_to_hard(const T *llrs, uint8_t *bits, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    constexpr size_t K = 16 / sizeof(T), SLICE = (size_t)1 << 38;
    const size_t pieces = bytes * 8 / K;
    for (size_t p0 = 0; p0 < pieces; p0 += SLICE) {
        const size_t np = pieces - p0 < SLICE ? pieces - p0 : SLICE;
        hipLaunchKernelGGL(llrs_to_hard_kernel<T>, dim3((unsigned)grid_for(np)), dim3(256), 0, stream, llrs + p0 * K, bits + p0 * K / 8, np);
    }
    return hipGetLastError();
}

#define LDPC_INSTANTIATE(T) \
    template hipError_t launch_hard_to_llrs<T>(const uint8_t *, T *, size_t, hipStream_t); \
    template hipError_t launch_llrs_to_hard<T>(const T *, uint8_t *, size_t, hipStream_t);
LDPC_INSTANTIATE(int8_t)
LDPC_INSTANTIATE(int16_t)
LDPC_INSTANTIATE(int32_t)
LDPC_INSTANTIATE(float)
LDPC_INSTANTIATE(double)
#undef LDPC_INSTANTIATE

}  // namespace ldpc
