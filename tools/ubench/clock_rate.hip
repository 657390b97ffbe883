// What does s_memtime count on gfx950, and at what rate under a VALU load?  One wave per SIMD .. four waves per SIMD spin
// on a dependent v_min/v_xor chain; s_memtime and s_memrealtime (100 MHz) are read at both ends.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_rate tools/ubench/clock_rate.hip && /tmp/clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) spin(unsigned long long *out, int loops, float seed)
{
    float a = threadIdx.x * 1.5f + seed, b = a * 3.f + 1.f, c = b - 7.f, d = a + b;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int l = 0; l < loops; ++l) {
        asm volatile("v_min_f32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_min_f32 %2, %3, %2\n v_xor_b32 %3, %0, %3\n"
                     "v_add_f32 %0, %1, %0\n v_mul_f32 %1, %2, %1\n v_min3_f32 %2, %3, %2, %0\n v_sub_f32 %3, %0, %3"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = m1 - m0; out[1] = r1 - r0; }
    if (a + b + c + d == 12345.f) out[2] = 1;
}
int main()
{
    unsigned long long *d, h[3];
    (void)hipMalloc(&d, 24);
    for (int threads : {256, 1024}) {
        for (int loops : {200000, 2000000}) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            spin<<<256, threads>>>(d, loops, 1.f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("%4d threads x 256 workgroups, %7d loops: %.3f ms by events | s_memtime %llu ticks (%.1f MHz) | s_memrealtime %llu ticks (%.1f MHz) | %.2f memtime ticks per 8-instruction loop\n",
                   threads, loops, ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), (double)h[0] / loops);
        }
    }
    return 0;
}
