// Micro-benchmark (round 6): issue cost of the instructions the f64 and i32 decode_ms kernels are made of, the way tools/ubench/valu_rate.hip
// measures the f32 ones: 256 workgroups x 1024 threads = 4 waves per SIMD on every CU, LOOPS x 32 copies of a group of independent
// instructions (inline asm), ns per wave-instruction per SIMD.  For the f64 / i32 rows of profiles/r06_final/rates_all_codes.txt: what is
// the issue-rate ceiling of decode_ms::<f64> / ::<i32> on gfx950, and which fraction of it do the kernels reach?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wide_rate tools/ubench/wide_rate.hip && /tmp/wide_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) \
    asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(e), "v"(f), "v"(ie), "v"(jf) : "vcc", "s20", "s21", "s22", "s23");
// operands: %0-%3 doubles a..d (rw), %4-%7 ints ia..id (rw), %8 %9 doubles e f, %10 %11 ints ie jf
template <int KIND>
__global__ void __launch_bounds__(1024) spin(double *out, int loops, double seed)
{
    double a = threadIdx.x * 1.5 + seed, b = a * 3. + 1., c = b - 7., d = a + b, e = 1.25, f = -0.75;
    int ia = threadIdx.x * 3 + 1, ib = ia * 5, ic = ib - 77, id = ia + ib, ie = 12345, jf = -777;
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) { BODY("v_add_f64 %0, %8, %1\n v_add_f64 %1, %9, %2\n v_add_f64 %2, %8, %3\n v_add_f64 %3, %9, %0") }
        if (KIND == 1) { BODY("v_mul_f64 %0, %8, %1\n v_mul_f64 %1, %9, %2\n v_mul_f64 %2, %8, %3\n v_mul_f64 %3, %9, %0") }
        if (KIND == 2) { BODY("v_fma_f64 %0, %8, %1, %9\n v_fma_f64 %1, %9, %2, %8\n v_fma_f64 %2, %8, %3, %9\n v_fma_f64 %3, %9, %0, %8") }
        if (KIND == 3) { BODY("v_min_f64 %0, %8, %1\n v_min_f64 %1, %9, %2\n v_min_f64 %2, %8, %3\n v_min_f64 %3, %9, %0") }
        if (KIND == 4) { BODY("v_min_f64 %0, |%8|, |%1|\n v_min_f64 %1, |%9|, |%2|\n v_min_f64 %2, |%8|, |%3|\n v_min_f64 %3, |%9|, |%0|") }
        if (KIND == 5) { BODY("v_cmp_lt_f64 vcc, %8, %1\n v_cmp_lt_f64 vcc, %9, %2\n v_cmp_lt_f64 vcc, %8, %3\n v_cmp_lt_f64 vcc, %9, %0") }
        // a double selected by a compare: v_cmp_f64 + two v_cndmask_b32 (there is no 64-bit select)
        if (KIND == 6) { BODY("v_cmp_lt_f64 vcc, %8, %1\n s_nop 1\n v_cndmask_b32 %4, 0, %5, vcc\n v_cndmask_b32 %6, 0, %7, vcc") }
        if (KIND == 7) { BODY("v_add_f64 %0, %8, %0\n v_add_f64 %0, %9, %0\n v_add_f64 %0, %8, %0\n v_add_f64 %0, %9, %0") }       // dependent chain
        if (KIND == 8) { BODY("v_add_f64 %0, %8, %0\n v_xor_b32 %4, %10, %4\n v_add_f64 %1, %9, %1\n v_xor_b32 %5, %11, %5") }     // f64 add / 32-bit F alternating
        if (KIND == 9) { BODY("v_add_f64 %0, %8, %0\n v_min_f64 %1, %9, %1\n v_add_f64 %2, %8, %2\n v_min_f64 %3, %9, %3") }      // add / min alternating
        if (KIND == 10) { BODY("v_max_f64 %0, %1, %1\n v_max_f64 %1, %2, %2\n v_max_f64 %2, %3, %3\n v_max_f64 %3, %0, %0") }     // canonicalise (x, x)
        // ---- i32 ----
        if (KIND == 20) { BODY("v_add_i32 %4, %10, %5 clamp\n v_add_i32 %5, %11, %6 clamp\n v_add_i32 %6, %10, %7 clamp\n v_add_i32 %7, %11, %4 clamp") }
        if (KIND == 21) { BODY("v_sub_i32 %4, %10, %5 clamp\n v_sub_i32 %5, %11, %6 clamp\n v_sub_i32 %6, %10, %7 clamp\n v_sub_i32 %7, %11, %4 clamp") }
        if (KIND == 22) { BODY("v_add_u32 %4, %10, %5\n v_add_u32 %5, %11, %6\n v_add_u32 %6, %10, %7\n v_add_u32 %7, %11, %4") }
        if (KIND == 23) { BODY("v_min3_i32 %4, %10, %5, %11\n v_min3_i32 %5, %11, %6, %10\n v_min3_i32 %6, %10, %7, %11\n v_min3_i32 %7, %11, %4, %10") }
        if (KIND == 24) { BODY("v_min_i32 %4, %10, %5\n v_min_i32 %5, %11, %6\n v_min_i32 %6, %10, %7\n v_min_i32 %7, %11, %4") }
        if (KIND == 25) { BODY("v_cmp_lt_i32 vcc, %10, %5\n v_cmp_lt_i32 vcc, %11, %6\n v_cmp_lt_i32 vcc, %10, %7\n v_cmp_lt_i32 vcc, %11, %4") }
        if (KIND == 26) { BODY("v_med3_i32 %4, %10, %5, %11\n v_med3_i32 %5, %11, %6, %10\n v_med3_i32 %6, %10, %7, %11\n v_med3_i32 %7, %11, %4, %10") }
        if (KIND == 27) { BODY("v_sub_i32 %4, 0, %5 clamp\n v_max_i32 %4, %4, %5\n v_sub_i32 %6, 0, %7 clamp\n v_max_i32 %6, %6, %7") }   // saturating_abs as the kernel forms it
        if (KIND == 28) { BODY("v_xad_u32 %4, %10, %5, %11\n v_xad_u32 %5, %11, %6, %10\n v_xad_u32 %6, %10, %7, %11\n v_xad_u32 %7, %11, %4, %10") }
        if (KIND == 29) { BODY("v_cvt_f32_i32 %4, %5\n v_cvt_f32_i32 %5, %6\n v_cvt_f32_i32 %6, %7\n v_cvt_f32_i32 %7, %4") }
        if (KIND == 30) { BODY("v_bfi_b32 %4, %10, %5, %11\n v_bfi_b32 %5, %11, %6, %10\n v_bfi_b32 %6, %10, %7, %11\n v_bfi_b32 %7, %11, %4, %10") }
    }
    if (a + b + c + d + (double)(ia + ib + ic + id) == 12345.) out[0] = a;
}
template <int KIND> void run(const char *name, int per_group = 4)
{
    double *d; (void)hipMalloc(&d, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256, threads = 1024;
    spin<KIND><<<blocks, threads>>>(d, 10, 1.);
    (void)hipEventRecord(a);
    spin<KIND><<<blocks, threads>>>(d, loops, 1.);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.3f ns per wave-instr per SIMD (%d per group)\n", name, ms * 1e6 / ((double)loops * 32 * per_group * 4), per_group);
    (void)hipFree(d);
}
int main()
{
    run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<3>("v_min_f64"); run<4>("v_min_f64 |abs|"); run<5>("v_cmp_lt_f64 vcc");
    run<6>("v_cmp_lt_f64 + 2 v_cndmask (one f64 select)", 3); run<7>("v_add_f64 dependent chain"); run<8>("v_add_f64 / v_xor_b32 alternating");
    run<9>("v_add_f64 / v_min_f64 alternating"); run<10>("v_max_f64 (x, x)");
    run<20>("v_add_i32 clamp"); run<21>("v_sub_i32 clamp"); run<22>("v_add_u32"); run<23>("v_min3_i32"); run<24>("v_min_i32"); run<25>("v_cmp_lt_i32 vcc");
    run<26>("v_med3_i32"); run<27>("saturating_abs: v_sub_i32 clamp + v_max_i32"); run<28>("v_xad_u32"); run<29>("v_cvt_f32_i32"); run<30>("v_bfi_b32");
    return 0;
}
