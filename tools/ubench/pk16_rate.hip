// Micro-benchmark: issue cost of the packed 16-bit VALU instructions on gfx950 (4 waves per SIMD, 256 CUs busy),
// to decide whether a packed-i16 message path for i8/i16 LLRs can beat the f32 pipeline.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/pk16_rate tools/ubench/pk16_rate.hip && /tmp/pk16_rate
// Each wave executes LOOPS x 32 x 4 instructions on four independent registers (inline asm); the last column is
// ns per wave-instruction per SIMD: ~0.9 = one instruction per 2 cycles (full rate), ~1.8 = one per 4 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) \
    asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(sm) : "vcc");
template <int KIND>
__global__ void __launch_bounds__(1024) spin(unsigned *out, int loops, unsigned seed)
{
    unsigned a = threadIdx.x * 0x10003u + seed, b = a * 3u + 1u, c = b - 7u, d = a + b, e = 0x00050003u, f = 0xfffd0002u;
    unsigned sm = 0x7fff7fffu;
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) { BODY("v_xor_b32 %0, %4, %1\n v_xor_b32 %1, %5, %2\n v_xor_b32 %2, %4, %3\n v_xor_b32 %3, %5, %0") }
        if (KIND == 1) { BODY("v_pk_add_i16 %0, %4, %1\n v_pk_add_i16 %1, %5, %2\n v_pk_add_i16 %2, %4, %3\n v_pk_add_i16 %3, %5, %0") }
        if (KIND == 2) { BODY("v_pk_add_i16 %0, %4, %1 clamp\n v_pk_add_i16 %1, %5, %2 clamp\n v_pk_add_i16 %2, %4, %3 clamp\n v_pk_add_i16 %3, %5, %0 clamp") }
        if (KIND == 3) { BODY("v_pk_sub_i16 %0, %4, %1 clamp\n v_pk_sub_i16 %1, %5, %2 clamp\n v_pk_sub_i16 %2, %4, %3 clamp\n v_pk_sub_i16 %3, %5, %0 clamp") }
        if (KIND == 4) { BODY("v_pk_min_i16 %0, %4, %1\n v_pk_min_i16 %1, %5, %2\n v_pk_min_i16 %2, %4, %3\n v_pk_min_i16 %3, %5, %0") }
        if (KIND == 5) { BODY("v_pk_max_i16 %0, %4, %1\n v_pk_max_i16 %1, %5, %2\n v_pk_max_i16 %2, %4, %3\n v_pk_max_i16 %3, %5, %0") }
        if (KIND == 6) { BODY("v_pk_min_u16 %0, %4, %1\n v_pk_min_u16 %1, %5, %2\n v_pk_min_u16 %2, %4, %3\n v_pk_min_u16 %3, %5, %0") }
        if (KIND == 7) { BODY("v_pk_ashrrev_i16 %0, 15, %1\n v_pk_ashrrev_i16 %1, 15, %2\n v_pk_ashrrev_i16 %2, 15, %3\n v_pk_ashrrev_i16 %3, 15, %0") }
        if (KIND == 8) { BODY("v_pk_lshrrev_b16 %0, 15, %1\n v_pk_lshrrev_b16 %1, 15, %2\n v_pk_lshrrev_b16 %2, 15, %3\n v_pk_lshrrev_b16 %3, 15, %0") }
        if (KIND == 9) { BODY("v_pk_mul_lo_u16 %0, %4, %1\n v_pk_mul_lo_u16 %1, %5, %2\n v_pk_mul_lo_u16 %2, %4, %3\n v_pk_mul_lo_u16 %3, %5, %0") }
        if (KIND == 10) { BODY("v_pk_mad_i16 %0, %4, %1, %5\n v_pk_mad_i16 %1, %5, %2, %4\n v_pk_mad_i16 %2, %4, %3, %5\n v_pk_mad_i16 %3, %5, %0, %4") }
        if (KIND == 11) { BODY("v_pk_add_f16 %0, %4, %1\n v_pk_add_f16 %1, %5, %2\n v_pk_add_f16 %2, %4, %3\n v_pk_add_f16 %3, %5, %0") }
        if (KIND == 12) { BODY("v_pk_min_f16 %0, %4, %1\n v_pk_min_f16 %1, %5, %2\n v_pk_min_f16 %2, %4, %3\n v_pk_min_f16 %3, %5, %0") }
        if (KIND == 13) { BODY("v_pk_mul_f16 %0, %4, %1\n v_pk_mul_f16 %1, %5, %2\n v_pk_mul_f16 %2, %4, %3\n v_pk_mul_f16 %3, %5, %0") }
        if (KIND == 14) { BODY("v_perm_b32 %0, %4, %1, %5\n v_perm_b32 %1, %5, %2, %4\n v_perm_b32 %2, %4, %3, %5\n v_perm_b32 %3, %5, %0, %4") }
        if (KIND == 15) { BODY("v_alignbit_b32 %0, %4, %1, 16\n v_alignbit_b32 %1, %5, %2, 16\n v_alignbit_b32 %2, %4, %3, 16\n v_alignbit_b32 %3, %5, %0, 16") }
        if (KIND == 16) { BODY("v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %2, %4 bitop3:0x96\n v_bitop3_b32 %2, %4, %3, %5 bitop3:0x96\n v_bitop3_b32 %3, %5, %0, %4 bitop3:0x96") }
        if (KIND == 17) { BODY("v_pk_sub_u16 %0, %1, %4 clamp\n v_pk_sub_u16 %1, %2, %5 clamp\n v_pk_sub_u16 %2, %3, %4 clamp\n v_pk_sub_u16 %3, %0, %5 clamp") }
        if (KIND == 18) { BODY("v_bfi_b32 %0, %6, %1, %4\n v_bfi_b32 %1, %6, %2, %5\n v_bfi_b32 %2, %6, %3, %4\n v_bfi_b32 %3, %6, %0, %5") }
        if (KIND == 19) { BODY("v_and_or_b32 %0, %4, %6, %1\n v_and_or_b32 %1, %5, %6, %2\n v_and_or_b32 %2, %4, %6, %3\n v_and_or_b32 %3, %5, %6, %0") }
        // mixes
        if (KIND == 30) { BODY("v_pk_add_i16 %0, %4, %0 clamp\n v_pk_min_i16 %1, %5, %1\n v_pk_add_i16 %2, %4, %2 clamp\n v_pk_min_i16 %3, %5, %3") }
        if (KIND == 31) { BODY("v_pk_add_i16 %0, %4, %0 clamp\n v_xor_b32 %1, %5, %1\n v_pk_add_i16 %2, %4, %2 clamp\n v_xor_b32 %3, %5, %3") }
        if (KIND == 32) { BODY("v_pk_min_i16 %0, %4, %0\n v_xor_b32 %1, %5, %1\n v_pk_min_i16 %2, %4, %2\n v_xor_b32 %3, %5, %3") }
        if (KIND == 33) { BODY("v_pk_min_u16 %0, %4, %0\n v_pk_ashrrev_i16 %1, 15, %1\n v_pk_min_u16 %2, %4, %2\n v_pk_ashrrev_i16 %3, 15, %3") }
        if (KIND == 34) { BODY("v_pk_min_f16 %0, %4, %0\n v_pk_add_f16 %1, %5, %1\n v_pk_min_f16 %2, %4, %2\n v_pk_add_f16 %3, %5, %3") }
        if (KIND == 35) { BODY("v_pk_min_i16 %0, %4, %0\n v_pk_max_i16 %1, %5, %1\n v_pk_min_i16 %2, %4, %2\n v_pk_max_i16 %3, %5, %3") }
        if (KIND == 36) { BODY("v_pk_mul_lo_u16 %0, %4, %0\n v_xor_b32 %1, %5, %1\n v_pk_mul_lo_u16 %2, %4, %2\n v_xor_b32 %3, %5, %3") }
        if (KIND == 37) { BODY("v_pk_min_i16 %0, %4, %0\n v_bitop3_b32 %1, %5, %1, %4 bitop3:0x96\n v_pk_min_i16 %2, %4, %2\n v_bitop3_b32 %3, %5, %3, %4 bitop3:0x96") }
    }
    if (a + b + c + d == 12345u) out[0] = a;
}
template <int KIND> void run(const char *name)
{
    unsigned *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256, threads = 1024;
    spin<KIND><<<blocks, threads>>>(d, 10, 1u);
    (void)hipEventRecord(a);
    spin<KIND><<<blocks, threads>>>(d, loops, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-40s %.3f ns per wave-instr per SIMD\n", name, ms * 1e6 / ((double)loops * 32 * 4 * 4));
}
int main()
{
    run<0>("v_xor_b32 (reference, full rate)");
    run<16>("v_bitop3_b32");
    run<1>("v_pk_add_i16"); run<2>("v_pk_add_i16 clamp"); run<3>("v_pk_sub_i16 clamp"); run<17>("v_pk_sub_u16 clamp");
    run<4>("v_pk_min_i16"); run<5>("v_pk_max_i16"); run<6>("v_pk_min_u16");
    run<7>("v_pk_ashrrev_i16"); run<8>("v_pk_lshrrev_b16"); run<9>("v_pk_mul_lo_u16"); run<10>("v_pk_mad_i16");
    run<11>("v_pk_add_f16"); run<12>("v_pk_min_f16"); run<13>("v_pk_mul_f16");
    run<14>("v_perm_b32"); run<15>("v_alignbit_b32"); run<18>("v_bfi_b32"); run<19>("v_and_or_b32");
    run<30>("pk_add_i16 clamp / pk_min_i16"); run<31>("pk_add_i16 clamp / xor"); run<32>("pk_min_i16 / xor");
    run<33>("pk_min_u16 / pk_ashrrev_i16"); run<34>("pk_min_f16 / pk_add_f16"); run<35>("pk_min_i16 / pk_max_i16");
    run<36>("pk_mul_lo_u16 / xor"); run<37>("pk_min_i16 / bitop3");
    return 0;
}
