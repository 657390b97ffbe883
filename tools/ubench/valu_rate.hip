// Micro-benchmark: VALU issue cost per instruction kind on gfx950 vs waves per SIMD.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
// Each wave executes LOOPS x 32 copies of one instruction on independent registers (inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) \
    asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(sm) : "vcc");
template <int KIND>
__global__ void __launch_bounds__(1024) spin(float *out, int loops, float seed, unsigned long long *clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x * 1.5f + seed, b = a * 3.f + 1.f, c = b - 7.f, d = a + b, e = 1.25f, f = -0.75f;
    unsigned sm = 0x7fffffffu;
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) { BODY("v_xor_b32 %0, %4, %1\n v_xor_b32 %1, %5, %2\n v_xor_b32 %2, %4, %3\n v_xor_b32 %3, %5, %0") }
        if (KIND == 1) { BODY("v_sub_f32 %0, %4, %1\n v_sub_f32 %1, %5, %2\n v_sub_f32 %2, %4, %3\n v_sub_f32 %3, %5, %0") }
        if (KIND == 2) { BODY("v_min3_f32 %0, %4, %1, %5\n v_min3_f32 %1, %5, %2, %4\n v_min3_f32 %2, %4, %3, %5\n v_min3_f32 %3, %5, %0, %4") }
        if (KIND == 3) { BODY("v_and_or_b32 %0, %4, %6, %1\n v_and_or_b32 %1, %5, %6, %2\n v_and_or_b32 %2, %4, %6, %3\n v_and_or_b32 %3, %5, %6, %0") }
        if (KIND == 4) { BODY("v_cmp_gt_i32 vcc, %4, %1\n v_cndmask_b32 %0, %1, %5, vcc\n v_cmp_ne_u32 vcc, %5, %3\n v_cndmask_b32 %2, %3, %4, vcc") }
        if (KIND == 5) { BODY("v_min_f32 %0, %4, %1\n v_min_f32 %1, %5, %2\n v_min_f32 %2, %4, %3\n v_min_f32 %3, %5, %0") }
        if (KIND == 6) { BODY("v_add_u32 %0, %4, %1\n v_add_u32 %1, %5, %2\n v_add_u32 %2, %4, %3\n v_add_u32 %3, %5, %0") }
        if (KIND == 8) { BODY("v_max_f32 %0, %4, %1\n v_max_f32 %1, %5, %2\n v_max_f32 %2, %4, %3\n v_max_f32 %3, %5, %0") }
        if (KIND == 9) { BODY("v_min_i32 %0, %4, %1\n v_min_i32 %1, %5, %2\n v_min_i32 %2, %4, %3\n v_min_i32 %3, %5, %0") }
        if (KIND == 10) { BODY("v_and_b32 %0, %4, %1\n v_or_b32 %1, %5, %2\n v_and_b32 %2, %4, %3\n v_or_b32 %3, %5, %0") }
        if (KIND == 11) { BODY("v_bfi_b32 %0, %6, %1, %5\n v_bfi_b32 %1, %6, %2, %4\n v_bfi_b32 %2, %6, %3, %5\n v_bfi_b32 %3, %6, %0, %4") }
        if (KIND == 12) { BODY("v_med3_f32 %0, %4, %1, %5\n v_med3_f32 %1, %5, %2, %4\n v_med3_f32 %2, %4, %3, %5\n v_med3_f32 %3, %5, %0, %4") }
        if (KIND == 13) { BODY("v_add_f32 %0, %4, %1\n v_add_f32 %1, %5, %2\n v_add_f32 %2, %4, %3\n v_add_f32 %3, %5, %0") }
        if (KIND == 14) { BODY("v_mul_f32 %0, %4, %1\n v_mul_f32 %1, %5, %2\n v_mul_f32 %2, %4, %3\n v_mul_f32 %3, %5, %0") }
        if (KIND == 15) { BODY("v_cmp_gt_i32 vcc, %4, %1\n v_cmp_gt_i32 vcc, %5, %2\n v_cmp_ne_u32 vcc, %5, %3\n v_cmp_lt_f32 vcc, %4, %0") }
        if (KIND == 16) { BODY("v_cndmask_b32 %0, %1, %5, vcc\n v_cndmask_b32 %1, %2, %4, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %5, vcc") }
        if (KIND == 17) { BODY("v_xor_b32_e64 %0, %4, %1\n v_xor_b32_e64 %1, %5, %2\n v_xor_b32_e64 %2, %4, %3\n v_xor_b32_e64 %3, %5, %0") }
        if (KIND == 7) { BODY("v_fma_f32 %0, %4, %1, %5\n v_fma_f32 %1, %5, %2, %4\n v_fma_f32 %2, %4, %3, %5\n v_fma_f32 %3, %5, %0, %4") }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && threadIdx.x == 0) clk[0] = t1 - t0;
    if (a + b + c + d == 12345.f) out[0] = a;
}
template <int KIND> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 4);
    unsigned long long *clk; (void)hipMalloc(&clk, 8);
    for (int threads : {256, 1024}) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        const int loops = 4000, blocks = 256;
        spin<KIND><<<blocks, threads>>>(d, 10, 1.f, clk);
        (void)hipEventRecord(a);
        spin<KIND><<<blocks, threads>>>(d, loops, 1.f, clk);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double instr_per_wave = (double)loops * 128;
        const double simd_waves = threads / 256.0;
        unsigned long long cyc = 0; (void)hipMemcpy(&cyc, clk, 8, hipMemcpyDeviceToHost);
        printf("%-22s waves/SIMD %.0f: %.3f ns, %.2f shader-cycles per wave-instr per SIMD (clock %.2f GHz)\n", name, simd_waves,
               ms * 1e6 / (instr_per_wave * simd_waves), (double)cyc / (instr_per_wave * simd_waves), (double)cyc / (ms * 1e6));
    }
}
int main()
{
    run<0>("v_xor_b32"); run<6>("v_add_u32"); run<1>("v_sub_f32"); run<5>("v_min_f32"); run<2>("v_min3_f32"); run<7>("v_fma_f32");
    run<3>("v_and_or_b32(sgpr)"); run<4>("v_cmp+v_cndmask");
    run<8>("v_max_f32"); run<9>("v_min_i32"); run<10>("v_and/v_or"); run<11>("v_bfi_b32"); run<12>("v_med3_f32");
    run<13>("v_add_f32"); run<14>("v_mul_f32"); run<15>("v_cmp x4"); run<16>("v_cndmask x4"); run<17>("v_xor_b32_e64");
    return 0;
}
