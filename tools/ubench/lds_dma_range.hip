// lds_dma_range.hip -- how far into a workgroup's LDS allocation can an LDS-DMA (global_load_lds_dwordx4, destination base in M0) write
// on gfx950?  (ds_write_addtid_b32 takes M0[15:0]: 64 KB.)  One wave DMAs 1 KiB to LDS offset X for several X up to 159 KB, waits
// (vmcnt(0)), reads it back with ds_read and compares; prints the verdict per X.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_range.hip -o build/ub/lds_dma_range && build/ub/lds_dma_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) probe(const uint32_t *src, uint32_t *out, uint32_t lds_off)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<uint32_t *>(lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t *g = src + lane * 4;                     // each lane's 16 bytes
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + lds_off);   // wave-uniform LDS byte address
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    __syncthreads();
    // where did the 1 KiB land?  report the first LDS dword that is no longer the fill pattern, and whether [lds_off, +1 KiB) == src
    uint32_t good = 1;
    for (int k = 0; k < 4; ++k) good &= reinterpret_cast<uint32_t *>(lds + lds_off)[lane * 4 + k] == src[lane * 4 + k];
    const unsigned long long all = __ballot(good);
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < 160 * 1024 / 4; i += 64)
        if (reinterpret_cast<uint32_t *>(lds)[i] != 0xDEADBEEFu && i * 4 < first) first = i * 4;
    for (int s = 32; s; s >>= 1) { const uint32_t o = __shfl_xor(first, s); first = o < first ? o : first; }
    if (lane == 0) { out[0] = all == ~0ull; out[1] = first; }
}

int main()
{
    uint32_t *src, *out;
    CK(hipMalloc(&src, 1024)); CK(hipMalloc(&out, 8));
    std::vector<uint32_t> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    CK(hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (uint32_t kb : {0u, 16u, 48u, 63u, 64u, 65u, 88u, 100u, 127u, 128u, 144u, 159u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, kb * 1024);
        CK(hipDeviceSynchronize());
        uint32_t r[2];
        CK(hipMemcpy(r, out, 8, hipMemcpyDeviceToHost));
        printf("LDS-DMA to offset %3u KB: %s (first changed LDS byte: %u)\n", kb, r[0] ? "landed there" : "NOT there", r[1]);
    }
    return 0;
}
