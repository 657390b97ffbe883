// hip_backend.hpp -- the gfx950 backend of the bit-sliced kernels (decode_ms_bitslice.hpp, decode_bf_bitslice.hpp): a wave register is
// a uint32_t per lane, the plane arithmetic is v_bitop3_b32, the lane permutation ds_bpermute_b32 + v_alignbit_b32.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef BS_FN
#define BS_FN __device__ __forceinline__
#endif

namespace ldpc {
namespace bs {

struct HipBackend {
    using V = uint32_t;
    char *lds;

    static BS_FN void fence() { __builtin_amdgcn_sched_barrier(0); }
    static BS_FN void mem_fence() { asm volatile("" ::: "memory"); }      // no memory access of this thread moves across (compiler only)
    static BS_FN void lds_wait() { __builtin_amdgcn_s_waitcnt(0xC07F); }   // s_waitcnt lgkmcnt(0): every LDS result of this wave is in its register
    static BS_FN void pin(V &x) { asm volatile("" : "+v"(x)); }          // the value exists in a register HERE (see the iteration)
    static BS_FN V c(uint32_t x) { return x; }
    BS_FN V lane() const { return threadIdx.x & 63u; }
    template <int TT> static BS_FN V bitop3(V a, V b, V cc) { return (V)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)cc, TT); }
    static BS_FN V and_(V a, V b) { return a & b; }
    static BS_FN V or_(V a, V b) { return a | b; }
    static BS_FN V xor_(V a, V b) { return a ^ b; }
    static BS_FN V andn(V a, V b) { return a & ~b; }
    static BS_FN V not_(V a) { return ~a; }
    static BS_FN V add(V a, V b) { return a + b; }
    static BS_FN V sub(V a, V b) { return a - b; }
    static BS_FN V mul_u(V a, uint32_t k) { return a * k; }
    static BS_FN V shl(V a, int s) { return a << s; }
    static BS_FN V shr(V a, int s) { return a >> s; }
    static BS_FN V sar(V a, int s) { return (V)((int32_t)a >> s); }
    static BS_FN V shl_v(V a, V s) { return a << (s & 31u); }
    static BS_FN V shr_v(V a, V s) { return a >> (s & 31u); }
    static BS_FN V bfe(V v, V off, int width) { return __builtin_amdgcn_ubfe(v, off, (uint32_t)width); }
    static BS_FN V rotr(V x, V amt) { return __builtin_amdgcn_alignbit(x, x, amt); }
    static BS_FN V less_u(V a, V b) { return a < b ? 0xFFFFFFFFu : 0u; }
    static BS_FN V eq(V a, V b) { return a == b ? 0xFFFFFFFFu : 0u; }
    // lane i takes x of lane (i & ~3) | bits 2i+1:2i of CTRL: v_mov_b32_dpp quad_perm (the identity costs nothing)
    template <int CTRL> static BS_FN V quad_perm(V x)
    {
        if constexpr (CTRL == 0xE4) return x;
        else return (V)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true);
    }
    BS_FN V bperm(V addr, V x) const { return (V)__builtin_amdgcn_ds_bpermute((int)addr, (int)x); }
    BS_FN V lds_read32(V addr) const { return *reinterpret_cast<const uint32_t *>(lds + addr); }
    BS_FN void lds_write32(V addr, V v) { *reinterpret_cast<uint32_t *>(lds + addr) = v; }
    BS_FN V lds_read_u8(V addr) const { return *reinterpret_cast<const uint8_t *>(lds + addr); }
    BS_FN V lds_read_u16(V addr) const { return *reinterpret_cast<const uint16_t *>(lds + addr); }
    BS_FN void lds_write16(V addr, V v) { *reinterpret_cast<uint16_t *>(lds + addr) = (uint16_t)v; }
    BS_FN void lds_write32_if(V addr, V v, V pred) { if (pred) *reinterpret_cast<uint32_t *>(lds + addr) = v; }
    static BS_FN V gload32(const void *p, V off, V pred)
    {
        return pred ? *reinterpret_cast<const uint32_t *>(static_cast<const char *>(p) + off) : 0u;
    }
    static BS_FN V gload32(const void *p, V off) { return *reinterpret_cast<const uint32_t *>(static_cast<const char *>(p) + off); }
    // 16 bytes per lane at a 4-byte-aligned address (global_load_dwordx4 takes any dword alignment).  The LLRs are read once: a
    // non-temporal stream, so that they do not push the few values the rate-1/2 kernels keep in scratch (11-14 spilled registers per
    // wave, re-used every iteration) out of the L2 -- with default-policy loads 28 % more bytes were fetched and 2.9 x the output bytes
    // written at the HBM boundary (TM8192 i8; profiles/r05_kbench/spill_leak.txt).  Output stores: non-temporal too in the kernels that spill
    // (gstore32_stream: the rate-1/2 codes), default policy in the others: a codeword's 4 * W bytes per block column are a partial line
    // where W < 32, which the L2 merges and a non-temporal store does not (TM5120: 1.5 x the output bytes for no gain in rate).
    static BS_FN void gload128(const void *p, V off, V (&w)[4])
    {
        typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(static_cast<const char *>(p) + off));
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
    BS_FN void lds_write128(V addr, const V (&w)[4])                    // 16-byte aligned
    {
        typedef uint32_t u4a __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u4a *>(lds + addr) = u4a{w[0], w[1], w[2], w[3]};
    }
    static BS_FN void gstore32(void *p, V off, V v, V pred) { if (pred) *reinterpret_cast<uint32_t *>(static_cast<char *>(p) + off) = v; }
    static BS_FN void gstore32_stream(void *p, V off, V v, V pred) { if (pred) __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(static_cast<char *>(p) + off)); }
    static BS_FN void gstore8(void *p, V off, V v, V pred) { if (pred) static_cast<uint8_t *>(p)[off] = (uint8_t)v; }
    // lane-wise select by a 64-bit lane mask: one v_cndmask_b32 with the mask in an SGPR pair, no plane of the mask in a register
    static BS_FN V select_lanes(uint64_t m, V a, V b)
    {
        V r;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
        return r;
    }
    static BS_FN uint64_t ballot(V x) { return __ballot(x != 0u); }
    static BS_FN uint32_t readlane(V x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }      // (l wave-uniform)
    BS_FN V plane_of(uint64_t m) const { return ((m >> (threadIdx.x & 63u)) & 1ull) ? 0xFFFFFFFFu : 0u; }
};

}  // namespace bs
}  // namespace ldpc
