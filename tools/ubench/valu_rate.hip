// Micro-benchmark: VALU issue cost per instruction kind on gfx950 (4 waves per SIMD, 256 CUs busy).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
// Each wave executes LOOPS x 32 x 4 copies of one instruction on independent registers (inline asm).
// Finding (round 1): plain VOP2 ALU ops (xor/and/or/add/sub/mul, f32 add/sub/mul, shifts) issue twice as
// fast as min/max, compares, v_cndmask and every 3-operand VOP3 form (min3/med3/fma/bfi/and_or).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
#define BODY(INS) \
    asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p) : "v"(e), "v"(f), "s"(sm), "v"(q) : "vcc", "s20", "s21", "s22", "s23");
// operand map: %0-%3 a,b,c,d  %4 p(pair rw)  -> shifted below
template <int KIND>
__global__ void __launch_bounds__(1024) spin(float *out, int loops, float seed)
{
    float a = threadIdx.x * 1.5f + seed, b = a * 3.f + 1.f, c = b - 7.f, d = a + b, e = 1.25f, f = -0.75f;
    float2_ p = {a, b}, q = {c, d};
    unsigned sm = 0x7fffffffu;
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) { BODY("v_xor_b32 %0, %5, %1\n v_xor_b32 %1, %6, %2\n v_xor_b32 %2, %5, %3\n v_xor_b32 %3, %6, %0") }
        if (KIND == 1) { BODY("v_sub_f32 %0, %5, %1\n v_sub_f32 %1, %6, %2\n v_sub_f32 %2, %5, %3\n v_sub_f32 %3, %6, %0") }
        if (KIND == 2) { BODY("v_mul_f32 %0, %5, %1\n v_mul_f32 %1, %6, %2\n v_mul_f32 %2, %5, %3\n v_mul_f32 %3, %6, %0") }
        if (KIND == 3) { BODY("v_min_f32 %0, %5, %1\n v_min_f32 %1, %6, %2\n v_min_f32 %2, %5, %3\n v_min_f32 %3, %6, %0") }
        if (KIND == 4) { BODY("v_min3_f32 %0, %5, %1, %6\n v_min3_f32 %1, %6, %2, %5\n v_min3_f32 %2, %5, %3, %6\n v_min3_f32 %3, %6, %0, %5") }
        if (KIND == 5) { BODY("v_ashrrev_i32 %0, 31, %1\n v_ashrrev_i32 %1, 31, %2\n v_ashrrev_i32 %2, 31, %3\n v_ashrrev_i32 %3, 31, %0") }
        if (KIND == 6) { BODY("v_lshlrev_b32 %0, 1, %1\n v_lshlrev_b32 %1, 1, %2\n v_lshlrev_b32 %2, 1, %3\n v_lshlrev_b32 %3, 1, %0") }
        if (KIND == 7) { BODY("v_not_b32 %0, %1\n v_not_b32 %1, %2\n v_not_b32 %2, %3\n v_not_b32 %3, %0") }
        if (KIND == 8) { BODY("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0") }
        if (KIND == 9) { BODY("v_sub_u32 %0, %5, %1\n v_sub_u32 %1, %6, %2\n v_sub_u32 %2, %5, %3\n v_sub_u32 %3, %6, %0") }
        if (KIND == 10) { BODY("v_max_i32 %0, %5, %1\n v_max_i32 %1, %6, %2\n v_max_i32 %2, %5, %3\n v_max_i32 %3, %6, %0") }
        if (KIND == 11) { BODY("v_min_u32 %0, %5, %1\n v_min_u32 %1, %6, %2\n v_min_u32 %2, %5, %3\n v_min_u32 %3, %6, %0") }
        if (KIND == 12) { BODY("v_bitop3_b32 %0, %5, %1, %6 bitop3:0x96\n v_bitop3_b32 %1, %6, %2, %5 bitop3:0x96\n v_bitop3_b32 %2, %5, %3, %6 bitop3:0x96\n v_bitop3_b32 %3, %6, %0, %5 bitop3:0x96") }
        if (KIND == 13) { BODY("v_or3_b32 %0, %5, %1, %6\n v_or3_b32 %1, %6, %2, %5\n v_or3_b32 %2, %5, %3, %6\n v_or3_b32 %3, %6, %0, %5") }
        if (KIND == 14) { BODY("v_add3_u32 %0, %5, %1, %6\n v_add3_u32 %1, %6, %2, %5\n v_add3_u32 %2, %5, %3, %6\n v_add3_u32 %3, %6, %0, %5") }
        if (KIND == 15) { BODY("v_xad_u32 %0, %5, %1, %6\n v_xad_u32 %1, %6, %2, %5\n v_xad_u32 %2, %5, %3, %6\n v_xad_u32 %3, %6, %0, %5") }
        if (KIND == 16) { BODY("v_lshl_add_u32 %0, %5, 2, %1\n v_lshl_add_u32 %1, %6, 2, %2\n v_lshl_add_u32 %2, %5, 2, %3\n v_lshl_add_u32 %3, %6, 2, %0") }
        if (KIND == 17) { BODY("v_mul_i32_i24 %0, %5, %1\n v_mul_i32_i24 %1, %6, %2\n v_mul_i32_i24 %2, %5, %3\n v_mul_i32_i24 %3, %6, %0") }
        if (KIND == 18) { BODY("v_mul_lo_u32 %0, %5, %1\n v_mul_lo_u32 %1, %6, %2\n v_mul_lo_u32 %2, %5, %3\n v_mul_lo_u32 %3, %6, %0") }
        if (KIND == 19) { BODY("v_cmp_lt_f32 vcc, %5, %1\n v_cmp_lt_f32 vcc, %6, %2\n v_cmp_lt_f32 vcc, %5, %3\n v_cmp_lt_f32 vcc, %6, %0") }
        if (KIND == 20) { BODY("v_cmp_class_f32 vcc, %5, %1\n v_cmp_class_f32 vcc, %6, %2\n v_cmp_class_f32 vcc, %5, %3\n v_cmp_class_f32 vcc, %6, %0") }
        if (KIND == 21) { BODY("v_cndmask_b32 %0, %5, %1, vcc\n v_cndmask_b32 %1, %6, %2, vcc\n v_cndmask_b32 %2, %5, %3, vcc\n v_cndmask_b32 %3, %6, %0, vcc") }
        if (KIND == 22) { BODY("v_sub_f32_e64 %0, |%5|, %1\n v_sub_f32_e64 %1, |%6|, %2\n v_sub_f32_e64 %2, |%5|, %3\n v_sub_f32_e64 %3, |%6|, %0") }
        if (KIND == 23) { BODY("v_add_f32 %0, %5, %1\n v_add_f32 %1, %6, %2\n v_add_f32 %2, %5, %3\n v_add_f32 %3, %6, %0") }
        if (KIND == 24) { BODY("v_max_f32 %0, %1, %1\n v_max_f32 %1, %2, %2\n v_max_f32 %2, %3, %3\n v_max_f32 %3, %0, %0") }
        if (KIND == 25) { BODY("v_med3_i32 %0, %5, %1, %6\n v_med3_i32 %1, %6, %2, %5\n v_med3_i32 %2, %5, %3, %6\n v_med3_i32 %3, %6, %0, %5") }
        if (KIND == 26) { BODY("v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %4, %8, %4") }
        if (KIND == 27) { BODY("v_pk_max_i16 %0, %5, %1\n v_pk_max_i16 %1, %6, %2\n v_pk_max_i16 %2, %5, %3\n v_pk_max_i16 %3, %6, %0") }
        if (KIND == 28) { BODY("v_pk_add_i16 %0, %5, %1\n v_pk_add_i16 %1, %6, %2\n v_pk_add_i16 %2, %5, %3\n v_pk_add_i16 %3, %6, %0") }
        if (KIND == 29) { BODY("v_and_b32 %0, %5, %1\n v_and_b32 %1, %6, %2\n v_and_b32 %2, %5, %3\n v_and_b32 %3, %6, %0") }
        // round 3: the pieces of the clamp form of the self-correction (Ops<float>::clamp_to_side)
        if (KIND == 60) { BODY("v_fmac_f32 %0, %5, %1\n v_fmac_f32 %1, %6, %2\n v_fmac_f32 %2, %5, %3\n v_fmac_f32 %3, %6, %0") }
        if (KIND == 61) { BODY("v_fma_f32 %0, %5, %1, %6\n v_fma_f32 %1, %6, %2, %5\n v_fma_f32 %2, %5, %3, %6\n v_fma_f32 %3, %6, %0, %5") }
        if (KIND == 62) { BODY("v_med3_f32 %0, %5, 0, %1\n v_med3_f32 %1, %6, 0, %2\n v_med3_f32 %2, %5, 0, %3\n v_med3_f32 %3, %6, 0, %0") }
        if (KIND == 63) { BODY("v_mul_legacy_f32_e64 %0, %5, %1\n v_mul_legacy_f32_e64 %1, %6, %2\n v_mul_legacy_f32_e64 %2, %5, %3\n v_mul_legacy_f32_e64 %3, %6, %0") }
        // the new edge update on four independent edges: sub, mov, fmac, med3 (against KIND 30 / 31: sub, bitop3, cmp, cndmask)
        if (KIND == 64) { BODY("v_sub_f32 %0, %5, %0\n v_mov_b32 %1, %0\n v_fmac_f32 %1, %7, %2\n v_med3_f32 %2, %0, 0, %1\n"
                               "v_sub_f32 %3, %6, %3\n v_mov_b32 %1, %3\n v_fmac_f32 %1, %7, %2\n v_med3_f32 %2, %3, 0, %1") }

        // ---- mixes of fast (F: v_xor) and slow (S: v_min_f32) independent ops: what does the order cost? ----
        if (KIND == 40) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_min_f32 %1, %6, %1\n v_min_f32 %3, %6, %3") }                 // FFSS
        if (KIND == 41) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_xor_b32 %0, %6, %0\n v_xor_b32 %2, %6, %2\n"
                               "v_min_f32 %1, %6, %1\n v_min_f32 %3, %6, %3\n v_min_f32 %1, %5, %1\n v_min_f32 %3, %5, %3") }                // FFFFSSSS
        if (KIND == 42) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_min_f32 %1, %6, %1") }                                          // FFS
        if (KIND == 43) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_xor_b32 %3, %6, %3\n v_min_f32 %1, %6, %1") }                // FFFS
        if (KIND == 44) { BODY("v_xor_b32 %0, %5, %0\n v_min_f32 %1, %6, %1\n v_min_f32 %3, %6, %3") }                                          // FSS
        if (KIND == 45) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %0, %6, %0\n v_xor_b32 %0, %5, %0\n v_xor_b32 %0, %6, %0") }                // F dependent chain
        if (KIND == 46) { BODY("v_min_f32 %1, %6, %1\n v_min_f32 %1, %5, %1\n v_min_f32 %1, %6, %1\n v_min_f32 %1, %5, %1") }                // S dependent chain
        if (KIND == 47) { BODY("v_cmp_ngt_f32 vcc, 0, %1\n v_xor_b32 %0, %5, %0\n v_cmp_ngt_f32 vcc, 0, %3\n v_xor_b32 %2, %5, %2") }        // cmp F cmp F
        if (KIND == 48) { BODY("v_cndmask_b32 %1, 0, %0, vcc\n v_xor_b32 %0, %5, %0\n v_cndmask_b32 %3, 0, %2, vcc\n v_xor_b32 %2, %5, %2") }  // cnd F cnd F
        if (KIND == 49) { BODY("v_cmp_ngt_f32 vcc, 0, %1\n s_nop 1\n v_cndmask_b32 %1, 0, %0, vcc\n v_cmp_ngt_f32 vcc, 0, %3\n s_nop 1\n v_cndmask_b32 %3, 0, %2, vcc") }  // cmp cnd pairs
        if (KIND == 50) { BODY("v_xor_b32 %0, %5, %0\n v_min3_f32 %1, %6, %1, %5\n v_xor_b32 %2, %5, %2\n v_min3_f32 %3, %6, %3, %5") }      // F min3 alternating
        if (KIND == 51) { BODY("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_min3_f32 %1, %6, %1, %5\n v_min3_f32 %3, %6, %3, %5") }      // FF min3 min3
        if (KIND == 52) { BODY("v_sub_f32 %0, %5, %0\n v_min_f32 %1, %6, %1\n v_sub_f32 %2, %5, %2\n v_min_f32 %3, %6, %3") }                // sub/min alternating
        if (KIND == 53) { BODY("v_bitop3_b32 %0, %5, %0, %6 bitop3:0x96\n v_min_f32 %1, %6, %1\n v_bitop3_b32 %2, %5, %2, %6 bitop3:0x96\n v_min_f32 %3, %6, %3") }  // bitop3/min alternating
        // ---- the self-correcting edge update of decode_ms (sub, bitop3, cmp, cndmask) on four independent edges ----
        // 30: as the compiler emits it (one edge after the other through VCC, hazard nops)
        if (KIND == 30) { BODY("v_sub_f32 %0, %5, %0\n v_bitop3_b32 %1, %1, %0, %7 bitop3:0x78\n v_cmp_ngt_f32 vcc, 0, %1\n s_nop 1\n v_cndmask_b32 %1, 0, %0, vcc\n"
                               "v_sub_f32 %2, %6, %2\n v_bitop3_b32 %3, %3, %2, %7 bitop3:0x78\n v_cmp_ngt_f32 vcc, 0, %3\n s_nop 1\n v_cndmask_b32 %3, 0, %2, vcc") }
        // 31: two edges interleaved, compares into separate SGPR pairs (no VCC chain, no nops)
        if (KIND == 31) { BODY("v_sub_f32 %0, %5, %0\n v_sub_f32 %2, %6, %2\n v_bitop3_b32 %1, %1, %0, %7 bitop3:0x78\n v_bitop3_b32 %3, %3, %2, %7 bitop3:0x78\n"
                               "v_cmp_ngt_f32_e64 s[20:21], 0, %1\n v_cmp_ngt_f32_e64 s[22:23], 0, %3\n v_cndmask_b32_e64 %1, 0, %0, s[20:21]\n v_cndmask_b32_e64 %3, 0, %2, s[22:23]") }
        // 32: like 30 without the nops but the second edge's sub/bitop3 between compare and select
        if (KIND == 32) { BODY("v_sub_f32 %0, %5, %0\n v_bitop3_b32 %1, %1, %0, %7 bitop3:0x78\n v_cmp_ngt_f32 vcc, 0, %1\n v_sub_f32 %2, %6, %2\n v_bitop3_b32 %3, %3, %2, %7 bitop3:0x78\n v_cndmask_b32 %1, 0, %0, vcc\n"
                               "v_cmp_ngt_f32 vcc, 0, %3\n s_nop 1\n v_cndmask_b32 %3, 0, %2, vcc") }
        // 33: alternating fast / slow independent ops (does the fast rate survive the mix?)
        if (KIND == 33) { BODY("v_xor_b32 %0, %5, %0\n v_min_f32 %1, %6, %1\n v_xor_b32 %2, %5, %2\n v_min_f32 %3, %6, %3") }
        // 34: min3 tree fragment with |abs| modifiers as in exclusive_min
        if (KIND == 34) { BODY("v_min3_f32 %0, |%5|, |%1|, %6\n v_min3_f32 %1, |%6|, |%2|, %5\n v_min3_f32 %2, |%5|, |%3|, %6\n v_min3_f32 %3, |%6|, |%0|, %5") }
    }
    if (a + b + c + d + p.x == 12345.f) out[0] = a;
}
template <int KIND> void run(const char *name, int per_group = 4)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256, threads = 1024;
    spin<KIND><<<blocks, threads>>>(d, 10, 1.f);
    (void)hipEventRecord(a);
    spin<KIND><<<blocks, threads>>>(d, loops, 1.f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s %.3f ns per wave-instr per SIMD (%d VALU per group)\n", name, ms * 1e6 / ((double)loops * 32 * per_group * 4), per_group);
}
int main()
{
    run<0>("v_xor_b32");
    run<1>("v_sub_f32");
    run<2>("v_mul_f32");
    run<3>("v_min_f32");
    run<4>("v_min3_f32");
    run<5>("v_ashrrev_i32");
    run<6>("v_lshlrev_b32");
    run<7>("v_not_b32");
    run<8>("v_mov_b32");
    run<9>("v_sub_u32");
    run<10>("v_max_i32");
    run<11>("v_min_u32");
    run<12>("v_bitop3_b32");
    run<13>("v_or3_b32");
    run<14>("v_add3_u32");
    run<15>("v_xad_u32");
    run<16>("v_lshl_add_u32");
    run<17>("v_mul_i32_i24");
    run<18>("v_mul_lo_u32");
    run<19>("v_cmp_lt_f32 vcc");
    run<20>("v_cmp_class_f32");
    run<21>("v_cndmask (vcc const)");
    run<22>("v_sub_f32 |abs| e64");
    run<23>("v_add_f32 sdwa?");
    run<24>("v_max_f32 (x,x)");
    run<25>("v_med3_i32");
    run<26>("v_pk_add_f32");
    run<27>("v_pk_min? v_pk_max_i16");
    run<28>("v_pk_add_i16");
    run<29>("v_and_b32");
    run<60>("v_fmac_f32 (VOP2)"); run<61>("v_fma_f32 (VOP3)"); run<62>("v_med3_f32 x, 0, y"); run<63>("v_mul_legacy_f32 (VOP3)");
    run<64>("edge update, clamp form", 8);

    run<40>("FFSS", 4); run<41>("FFFFSSSS", 8); run<42>("FFS", 3); run<43>("FFFS", 4); run<44>("FSS", 3);
    run<45>("F dependent chain", 4); run<46>("S dependent chain", 4); run<47>("cmp F cmp F", 4); run<48>("cnd F cnd F", 4);
    run<49>("cmp cnd pairs", 4); run<50>("F min3 alternating", 4); run<51>("FF min3 min3", 4); run<52>("sub/min alternating", 4);
    run<53>("bitop3/min alternating", 4);
    run<30>("edge update, serial via VCC", 8);
    run<31>("edge update, 2 interleaved, SGPR", 8);
    run<32>("edge update, overlapped VCC", 8);
    run<33>("xor/min alternating", 4);
    run<34>("v_min3_f32 |abs|", 4);
    return 0;
}
