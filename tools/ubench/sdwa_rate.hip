// Micro-benchmark (round 4): can the integer LLR types keep a genuinely narrow message state on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sdwa_rate tools/ubench/sdwa_rate.hip && /tmp/sdwa_rate
// Part 1 -- SDWA (byte / word operand selects on VOP1/VOP2, partial destination writes with UNUSED_PRESERVE):
//   issue cost at 4 (and 2, 1) waves per SIMD of the operations a byte-packed min-sum update needs, on independent
//   registers, on the four bytes of ONE register (the read-modify-write dependency of a partial write), and as the
//   dependent add -> min -> max chain of a saturating i8 add; beside them the VOP3 16-bit forms (v_add_i16 clamp with
//   op_sel, v_med3_i16, v_min3_i16) and the f32 instructions the shipped kernels use for the same work.
// Part 2 -- the primitives of a BIT-SLICED message state (32 indices of a block per register, one register per bit plane):
//   v_bitop3_b32 ripple chains, v_alignbit_b32 with a VGPR shift, ds_bpermute_b32, DPP row / wave rotations.
// Same harness as valu_rate.hip: every wave executes LOOPS x 32 groups of the listed instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) \
    asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");
// operands: %0-%3 a b c d (read-write)   %4 %5 e f (read-only values)   %6 g (read-only: lane address / shift)   %7 h
#define P " dst_unused:UNUSED_PRESERVE "
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS) spin(unsigned *out, int loops, unsigned seed)
{
    unsigned a = threadIdx.x * 0x01030507u + seed, b = a * 3u + 1u, c = b - 7u, d = a + b, e = 0x11223344u ^ seed, f = 0x05060708u + seed;
    unsigned g = ((threadIdx.x + 5) & 63) * 4, h = (threadIdx.x * 7) & 31;
    for (int l = 0; l < loops; ++l) {
        // ---- references ----
        if (KIND == 0) { BODY("v_xor_b32 %0, %4, %1\n v_xor_b32 %1, %5, %2\n v_xor_b32 %2, %4, %3\n v_xor_b32 %3, %5, %0") }
        if (KIND == 1) { BODY("v_add_f32 %0, %4, %1\n v_add_f32 %1, %5, %2\n v_add_f32 %2, %4, %3\n v_add_f32 %3, %5, %0") }
        if (KIND == 2) { BODY("v_med3_f32 %0, %4, %1, %5\n v_med3_f32 %1, %5, %2, %4\n v_med3_f32 %2, %4, %3, %5\n v_med3_f32 %3, %5, %0, %4") }
        if (KIND == 3) { BODY("v_min_i32 %0, %4, %1\n v_min_i32 %1, %5, %2\n v_min_i32 %2, %4, %3\n v_min_i32 %3, %5, %0") }
        if (KIND == 4) { BODY("v_add_u16 %0, %4, %1\n v_add_u16 %1, %5, %2\n v_add_u16 %2, %4, %3\n v_add_u16 %3, %5, %0") }
        if (KIND == 5) { BODY("v_min_i16 %0, %4, %1\n v_min_i16 %1, %5, %2\n v_min_i16 %2, %4, %3\n v_min_i16 %3, %5, %0") }
        if (KIND == 6) { BODY("v_max_i16 %0, %4, %1\n v_max_i16 %1, %5, %2\n v_max_i16 %2, %4, %3\n v_max_i16 %3, %5, %0") }
        // ---- SDWA, source selects only (full dword destination) ----
        if (KIND == 10) { BODY("v_add_u16_sdwa %0, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_add_u16_sdwa %1, %5, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
                                "v_add_u16_sdwa %2, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3\n v_add_u16_sdwa %3, %5, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0") }
        if (KIND == 11) { BODY("v_add_u16_sdwa %0, sext(%4), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_add_u16_sdwa %1, sext(%5), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
                                "v_add_u16_sdwa %2, sext(%4), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3\n v_add_u16_sdwa %3, sext(%5), sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0") }
        if (KIND == 12) { BODY("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                                "v_cvt_f32_i32_sdwa %2, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %3, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3") }
        // ---- SDWA, partial destination write (UNUSED_PRESERVE), independent registers ----
        if (KIND == 20) { BODY("v_add_u16_sdwa %0, %4, %1 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u16_sdwa %1, %5, %2 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_add_u16_sdwa %2, %4, %3 dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_add_u16_sdwa %3, %5, %0 dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        if (KIND == 21) { BODY("v_sub_u16_sdwa %0, %4, %1 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_sub_u16_sdwa %1, %5, %2 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_sub_u16_sdwa %2, %4, %3 dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_sub_u16_sdwa %3, %5, %0 dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        if (KIND == 22) { BODY("v_min_i16_sdwa %0, sext(%4), sext(%1) dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_min_i16_sdwa %1, sext(%5), sext(%2) dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_min_i16_sdwa %2, sext(%4), sext(%3) dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_min_i16_sdwa %3, sext(%5), sext(%0) dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        if (KIND == 23) { BODY("v_max_i16_sdwa %0, sext(%4), sext(%1) dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_max_i16_sdwa %1, sext(%5), sext(%2) dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_max_i16_sdwa %2, sext(%4), sext(%3) dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_max_i16_sdwa %3, sext(%5), sext(%0) dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        if (KIND == 24) { BODY("v_xor_b32_sdwa %0, %4, %1 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_xor_b32_sdwa %1, %5, %2 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_xor_b32_sdwa %2, %4, %3 dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_xor_b32_sdwa %3, %5, %0 dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        if (KIND == 25) { BODY("v_add_u16_sdwa %0, %4, %1 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_add_u16_sdwa %1, %5, %2 dst_sel:WORD_1" P "src0_sel:WORD_1 src1_sel:WORD_1\n"
                                "v_add_u16_sdwa %2, %4, %3 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_add_u16_sdwa %3, %5, %0 dst_sel:WORD_1" P "src0_sel:WORD_1 src1_sel:WORD_1") }
        if (KIND == 26) { BODY("v_min_i16_sdwa %0, %4, %1 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_min_i16_sdwa %1, %5, %2 dst_sel:WORD_1" P "src0_sel:WORD_1 src1_sel:WORD_1\n"
                                "v_min_i16_sdwa %2, %4, %3 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_min_i16_sdwa %3, %5, %0 dst_sel:WORD_1" P "src0_sel:WORD_1 src1_sel:WORD_1") }
        // ---- SDWA partial writes to the four bytes of ONE register (the packed layout): RMW dependency through the register ----
        if (KIND == 30) { BODY("v_add_u16_sdwa %0, %4, %0 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u16_sdwa %0, %5, %0 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_add_u16_sdwa %0, %4, %0 dst_sel:BYTE_2" P "src0_sel:BYTE_2 src1_sel:BYTE_2\n v_add_u16_sdwa %0, %5, %0 dst_sel:BYTE_3" P "src0_sel:BYTE_3 src1_sel:BYTE_3") }
        // four registers, each getting its four bytes in turn, interleaved so that consecutive writes go to different registers
        if (KIND == 31) { BODY("v_add_u16_sdwa %0, %4, %0 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u16_sdwa %1, %5, %1 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n"
                                "v_add_u16_sdwa %2, %4, %2 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u16_sdwa %3, %5, %3 dst_sel:BYTE_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n"
                                "v_add_u16_sdwa %0, %4, %0 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u16_sdwa %1, %5, %1 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                                "v_add_u16_sdwa %2, %4, %2 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u16_sdwa %3, %5, %3 dst_sel:BYTE_1" P "src0_sel:BYTE_1 src1_sel:BYTE_1") }
        // ---- the saturating i8 add of a packed layout: add (16-bit, sign-extended bytes) -> min 127 -> max -128, dependent, byte destination ----
        if (KIND == 40) { BODY("v_add_u16_sdwa %0, sext(%4), sext(%0) dst_sel:WORD_0" P "src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u16_sdwa %1, sext(%5), sext(%1) dst_sel:WORD_0" P "src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                                "v_min_i16_sdwa %0, %0, %6 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_min_i16_sdwa %1, %1, %6 dst_sel:WORD_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n"
                                "v_max_i16_sdwa %0, %0, %7 dst_sel:BYTE_0" P "src0_sel:WORD_0 src1_sel:WORD_0\n v_max_i16_sdwa %1, %1, %7 dst_sel:BYTE_0" P "src0_sel:WORD_0 src1_sel:WORD_0") }
        // the same work on the f32 pipe, as the shipped kernels do it: add + med3 (two edges)
        if (KIND == 41) { BODY("v_add_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_med3_f32 %0, %0, %6, %7\n v_med3_f32 %1, %1, %6, %7") }
        // VOP3 16-bit forms
        if (KIND == 50) { BODY("v_add_i16 %0, %4, %1 clamp\n v_add_i16 %1, %5, %2 clamp\n v_add_i16 %2, %4, %3 clamp\n v_add_i16 %3, %5, %0 clamp") }
        if (KIND == 51) { BODY("v_add_i16 %0, %4, %1 op_sel:[1,1,1] clamp\n v_add_i16 %1, %5, %2 op_sel:[1,1,1] clamp\n v_add_i16 %2, %4, %3 op_sel:[1,1,1] clamp\n v_add_i16 %3, %5, %0 op_sel:[1,1,1] clamp") }
        if (KIND == 52) { BODY("v_med3_i16 %0, %4, %1, %5\n v_med3_i16 %1, %5, %2, %4\n v_med3_i16 %2, %4, %3, %5\n v_med3_i16 %3, %5, %0, %4") }
        if (KIND == 53) { BODY("v_min3_i16 %0, %4, %1, %5\n v_min3_i16 %1, %5, %2, %4\n v_min3_i16 %2, %4, %3, %5\n v_min3_i16 %3, %5, %0, %4") }
        if (KIND == 54) { BODY("v_pk_add_i16 %0, %4, %1 clamp\n v_pk_add_i16 %1, %5, %2 clamp\n v_pk_add_i16 %2, %4, %3 clamp\n v_pk_add_i16 %3, %5, %0 clamp") }
        if (KIND == 55) { BODY("v_perm_b32 %0, %4, %1, %5\n v_perm_b32 %1, %5, %2, %4\n v_perm_b32 %2, %4, %3, %5\n v_perm_b32 %3, %5, %0, %4") }
        if (KIND == 56) { BODY("v_bfe_i32 %0, %1, 8, 8\n v_bfe_i32 %1, %2, 16, 8\n v_bfe_i32 %2, %3, 8, 8\n v_bfe_i32 %3, %0, 16, 8") }
        // ---- bit-sliced primitives ----
        // full adder, independent bits: sum = a ^ b ^ c (0x96), carry = majority (0xe8)
        if (KIND == 60) { BODY("v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %2, %4 bitop3:0xe8\n v_bitop3_b32 %2, %4, %3, %5 bitop3:0x96\n v_bitop3_b32 %3, %5, %0, %4 bitop3:0xe8") }
        // ripple: every instruction depends on the previous one's result (carry chain)
        if (KIND == 61) { BODY("v_bitop3_b32 %0, %4, %5, %0 bitop3:0xe8\n v_bitop3_b32 %0, %5, %4, %0 bitop3:0xe8\n v_bitop3_b32 %0, %4, %5, %0 bitop3:0xe8\n v_bitop3_b32 %0, %5, %4, %0 bitop3:0xe8") }
        // two independent ripple chains interleaved
        if (KIND == 62) { BODY("v_bitop3_b32 %0, %4, %5, %0 bitop3:0xe8\n v_bitop3_b32 %1, %5, %4, %1 bitop3:0xe8\n v_bitop3_b32 %0, %4, %5, %0 bitop3:0xe8\n v_bitop3_b32 %1, %5, %4, %1 bitop3:0xe8") }
        if (KIND == 63) { BODY("v_alignbit_b32 %0, %4, %1, %7\n v_alignbit_b32 %1, %5, %2, %7\n v_alignbit_b32 %2, %4, %3, %7\n v_alignbit_b32 %3, %5, %0, %7") }
        if (KIND == 64) { BODY("v_alignbit_b32 %0, %4, %1, 13\n v_alignbit_b32 %1, %5, %2, 13\n v_alignbit_b32 %2, %4, %3, 13\n v_alignbit_b32 %3, %5, %0, 13") }
        if (KIND == 65) { BODY("ds_bpermute_b32 %0, %6, %1\n ds_bpermute_b32 %1, %6, %2\n ds_bpermute_b32 %2, %6, %3\n ds_bpermute_b32 %3, %6, %0\n s_waitcnt lgkmcnt(0)") }
        if (KIND == 66) { BODY("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:1 row_mask:0xf bank_mask:0xf") }
        if (KIND == 67) { BODY("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_ror:1 row_mask:0xf bank_mask:0xf") }
        if (KIND == 68) { BODY("v_xor_b32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %1, %2, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %2, %3, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %3, %0, %3 row_ror:1 row_mask:0xf bank_mask:0xf") }
        // bpermute with the VALU work of a bit-sliced update in its shadow: 1 bpermute per 8 bitop3
        if (KIND == 69) { BODY("ds_bpermute_b32 %3, %6, %2\n v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %0, %4 bitop3:0xe8\n v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %0, %4 bitop3:0xe8\n"
                                "v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %0, %4 bitop3:0xe8\n v_bitop3_b32 %0, %4, %1, %5 bitop3:0x96\n v_bitop3_b32 %1, %5, %0, %4 bitop3:0xe8\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %2, %3, %2") }
        if (KIND == 70) { BODY("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %3, %0") }
        if (KIND == 71) { BODY("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %3, %0") }
    }
    if (a + b + c + d == 12345u) out[0] = a;
}
template <int KIND, int THREADS = 1024> void run(const char *name, int per_group = 4)
{
    unsigned *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256;
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, 10, 1u);
    (void)hipEventRecord(a);
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, loops, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const int waves_per_simd = THREADS / 256;
    printf("%-58s %d w/SIMD  %.3f ns per wave-instr per SIMD\n", name, waves_per_simd, ms * 1e6 / ((double)loops * 32 * per_group * waves_per_simd));
    (void)hipFree(d);
}
#define ALL3(K, NAME, G) run<K, 1024>(NAME, G); run<K, 512>(NAME, G); run<K, 256>(NAME, G);
int main()
{
    puts("-- references");
    ALL3(0, "v_xor_b32", 4) ALL3(1, "v_add_f32", 4) ALL3(2, "v_med3_f32", 4) ALL3(3, "v_min_i32", 4)
    ALL3(4, "v_add_u16 (VOP2)", 4) ALL3(5, "v_min_i16 (VOP2)", 4) ALL3(6, "v_max_i16 (VOP2)", 4)
    puts("-- SDWA: source selects, dword destination");
    ALL3(10, "v_add_u16_sdwa src BYTE_n", 4) ALL3(11, "v_add_u16_sdwa sext(src BYTE_n)", 4) ALL3(12, "v_cvt_f32_i32_sdwa sext(BYTE_n)", 4)
    puts("-- SDWA: byte / word destination, UNUSED_PRESERVE, independent registers");
    ALL3(20, "v_add_u16_sdwa dst BYTE_n preserve", 4) ALL3(21, "v_sub_u16_sdwa dst BYTE_n preserve", 4) ALL3(22, "v_min_i16_sdwa sext dst BYTE_n preserve", 4)
    ALL3(23, "v_max_i16_sdwa sext dst BYTE_n preserve", 4) ALL3(24, "v_xor_b32_sdwa dst BYTE_n preserve", 4) ALL3(25, "v_add_u16_sdwa dst WORD_n preserve", 4)
    ALL3(26, "v_min_i16_sdwa dst WORD_n preserve", 4)
    puts("-- SDWA: the four bytes of one register in turn (RMW through the register)");
    ALL3(30, "v_add_u16_sdwa bytes 0..3 of ONE register", 4) ALL3(31, "v_add_u16_sdwa 4 registers round-robin", 8)
    puts("-- saturating i8 add, two edges: SDWA add+min+max (6 instr) against f32 add+med3 (4 instr); ns PER EDGE = ns x instr / 2");
    ALL3(40, "sat add, SDWA chain (6 instr / 2 edges)", 6) ALL3(41, "sat add, f32 add+med3 (4 instr / 2 edges)", 4)
    puts("-- VOP3 16-bit forms");
    ALL3(50, "v_add_i16 clamp", 4) ALL3(51, "v_add_i16 clamp op_sel hi", 4) ALL3(52, "v_med3_i16", 4) ALL3(53, "v_min3_i16", 4) ALL3(54, "v_pk_add_i16 clamp", 4)
    ALL3(55, "v_perm_b32", 4) ALL3(56, "v_bfe_i32", 4)
    puts("-- bit-sliced primitives");
    ALL3(60, "v_bitop3 full adder, independent", 4) ALL3(61, "v_bitop3 ripple, one dependent chain", 4) ALL3(62, "v_bitop3 ripple, two chains interleaved", 4)
    ALL3(63, "v_alignbit_b32 (VGPR shift)", 4) ALL3(64, "v_alignbit_b32 (literal shift)", 4)
    ALL3(65, "ds_bpermute_b32 (4 + waitcnt)", 4) ALL3(66, "v_mov_b32_dpp row_ror:1", 4) ALL3(67, "v_mov_b32_dpp wave_ror:1", 4) ALL3(68, "v_xor_b32_dpp row_ror:1", 4)
    ALL3(69, "1 bpermute under 8 bitop3 + xor (10 instr)", 10)
    ALL3(70, "v_permlane32_swap", 4) ALL3(71, "v_permlane16_swap", 4)
    return 0;
}
