// Micro-benchmark (round 4, review item 5): what could the matrix cores do for the encoder?  parity = data [B x k] . G [k x (n-k)] mod 2 is
// the one dense contraction of the crate (/root/reference/src/encoder.rs:41-82).  The shipped encoder is bit-packed VALU work: one
// v_bitop3_b32 folds 32 data bits x 64 lanes = 2048 bit-MACs.  Here: issue rate of the MFMA forms that could hold 0/1 operands exactly --
// v_mfma_i32_32x32x32_i8 (32 x 32 x 32 = 32 768 MACs per wave instruction) and v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands
// (65 536 MACs) -- against v_bitop3_b32, all at 4 waves per SIMD on independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/ubench/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) spin(float *out, int loops)
{
    v4i a4 = {(int)threadIdx.x, 1, 2, 3}, b4 = {4, 5, 6, (int)threadIdx.x};
    v8i a8 = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {4, 5, 6, 7, 8, 9, 10, (int)threadIdx.x};
    v16i ci0 = {}, ci1 = {};
    v16f cf0 = {}, cf1 = {};
    unsigned x = threadIdx.x, y = 77, z = 99;
    for (int l = 0; l < loops; ++l) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) { ci0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ci0, 0, 0, 0); ci1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b4, a4, ci1, 0, 0, 0); }
            if (KIND == 1) {   // fp4 x fp4 (format code 4), unit scales
                cf0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, cf0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                cf1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, cf1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            if (KIND == 2) {   // fp8 x fp8 (format code 0)
                cf0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, cf0, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                cf1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, cf1, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            if (KIND == 3) { x = __builtin_amdgcn_bitop3_b32(x, y, z, 0x78); y = __builtin_amdgcn_bitop3_b32(y, z, x, 0x78); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += ci0[i] + ci1[i] + cf0[i] + cf1[i];
    if (s + x + y == 12345.f) out[0] = s;
}
template <int KIND> void run(const char *name, double macs_per_instr)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 4000, blocks = 256 * 4;              // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    spin<KIND><<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a);
    spin<KIND><<<blocks, 256>>>(d, loops);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)loops * 16 * 4;  // 16 instructions per loop, 4 waves per SIMD
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-44s %.2f ns per wave-instruction per SIMD   %.1f T MAC/s on 256 CUs\n", name, ns, macs_per_instr / ns * 1024 / 1e3);
}
int main()
{
    run<0>("v_mfma_i32_32x32x32_i8", 32768.0);
    run<1>("v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4)", 65536.0);
    run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8)", 65536.0);
    run<3>("v_bitop3_b32 (2048 bit-MACs: acc ^= g & d)", 2048.0);
    return 0;
}
