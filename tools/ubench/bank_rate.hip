// Micro-benchmark (round 4): does the VGPR bank of a three-source VALU instruction's operands change its issue cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bank_rate tools/ubench/bank_rate.hip && /tmp/bank_rate
// v_bitop3_b32 with explicit registers: the three sources in three different banks (register number mod 4), two in one bank, all
// three in one bank; the destination in the sources' bank or not; the same for a dependent ripple (sum / carry of a full adder, the
// shape of the bit-sliced decoder's saturating add).  Same timing harness as sdwa_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS) spin(unsigned *out, int loops)
{
    for (int l = 0; l < loops; ++l) {
        // independent: four destinations v16..v19, sources never written
        if (KIND == 0) { BODY("v_bitop3_b32 v16, v0, v1, v2 bitop3:0x96\n v_bitop3_b32 v17, v1, v2, v3 bitop3:0xe8\n v_bitop3_b32 v18, v2, v3, v0 bitop3:0x96\n v_bitop3_b32 v19, v3, v0, v1 bitop3:0xe8") }   // banks all different
        if (KIND == 1) { BODY("v_bitop3_b32 v16, v0, v4, v1 bitop3:0x96\n v_bitop3_b32 v17, v1, v5, v2 bitop3:0xe8\n v_bitop3_b32 v18, v2, v6, v3 bitop3:0x96\n v_bitop3_b32 v19, v3, v7, v0 bitop3:0xe8") }   // two sources share a bank
        if (KIND == 2) { BODY("v_bitop3_b32 v16, v0, v4, v8 bitop3:0x96\n v_bitop3_b32 v17, v1, v5, v9 bitop3:0xe8\n v_bitop3_b32 v18, v2, v6, v10 bitop3:0x96\n v_bitop3_b32 v19, v3, v7, v11 bitop3:0xe8") } // all three share a bank
        if (KIND == 3) { BODY("v_bitop3_b32 v16, v0, v0, v0 bitop3:0x96\n v_bitop3_b32 v17, v1, v1, v1 bitop3:0xe8\n v_bitop3_b32 v18, v2, v2, v2 bitop3:0x96\n v_bitop3_b32 v19, v3, v3, v3 bitop3:0xe8") }   // one register three times
        // two-source reference
        if (KIND == 4) { BODY("v_xor_b32 v16, v0, v1\n v_xor_b32 v17, v1, v2\n v_xor_b32 v18, v2, v3\n v_xor_b32 v19, v3, v0") }
        if (KIND == 5) { BODY("v_xor_b32 v16, v0, v4\n v_xor_b32 v17, v1, v5\n v_xor_b32 v18, v2, v6\n v_xor_b32 v19, v3, v7") }               // both sources in one bank
        // full-adder ripple (sum into v16.., carry chained through v20/v21), sources in different / equal banks
        if (KIND == 6) { BODY("v_bitop3_b32 v16, v0, v1, v20 bitop3:0x96\n v_bitop3_b32 v21, v0, v1, v20 bitop3:0xe8\n v_bitop3_b32 v17, v2, v3, v21 bitop3:0x96\n v_bitop3_b32 v20, v2, v3, v21 bitop3:0xe8") }   // a, x, c in banks 0 1 0 / 2 3 1
        if (KIND == 7) { BODY("v_bitop3_b32 v16, v0, v4, v20 bitop3:0x96\n v_bitop3_b32 v21, v0, v4, v20 bitop3:0xe8\n v_bitop3_b32 v17, v1, v5, v21 bitop3:0x96\n v_bitop3_b32 v20, v1, v5, v21 bitop3:0xe8") }   // a, x same bank, carry in that bank too (v20: 0, v21: 1)
        if (KIND == 8) { BODY("v_bitop3_b32 v16, v0, v5, v22 bitop3:0x96\n v_bitop3_b32 v23, v0, v5, v22 bitop3:0xe8\n v_bitop3_b32 v17, v1, v6, v23 bitop3:0x96\n v_bitop3_b32 v22, v1, v6, v23 bitop3:0xe8") }   // banks 0 1 2 / 1 2 3: all different
        // mux with a shared select (the saturation / minimum updates): sel, a, b
        if (KIND == 9) { BODY("v_bitop3_b32 v16, v0, v1, v2 bitop3:0xca\n v_bitop3_b32 v17, v0, v5, v6 bitop3:0xca\n v_bitop3_b32 v18, v0, v9, v10 bitop3:0xca\n v_bitop3_b32 v19, v0, v13, v14 bitop3:0xca") }
        if (KIND == 10) { BODY("v_bitop3_b32 v16, v0, v4, v8 bitop3:0xca\n v_bitop3_b32 v17, v0, v12, v4 bitop3:0xca\n v_bitop3_b32 v18, v0, v8, v12 bitop3:0xca\n v_bitop3_b32 v19, v0, v4, v12 bitop3:0xca") }
    }
    if (loops < 0) out[0] = 1;
}
template <int KIND, int THREADS = 1024> void run(const char *name)
{
    unsigned *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256;
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, 10);
    (void)hipEventRecord(a);
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, loops);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const int waves_per_simd = THREADS / 256;
    printf("%-72s %d w/SIMD  %.3f ns per wave-instr per SIMD\n", name, waves_per_simd, ms * 1e6 / ((double)loops * 32 * 4 * waves_per_simd));
    (void)hipFree(d);
}
#define ALL3(K, NAME) run<K, 1024>(NAME); run<K, 512>(NAME); run<K, 256>(NAME);
int main()
{
    ALL3(0, "v_bitop3 independent, sources in three banks")
    ALL3(1, "v_bitop3 independent, two sources in one bank")
    ALL3(2, "v_bitop3 independent, three sources in one bank")
    ALL3(3, "v_bitop3 independent, one register three times")
    ALL3(4, "v_xor_b32, sources in two banks")
    ALL3(5, "v_xor_b32, both sources in one bank")
    ALL3(6, "full-adder ripple, mixed banks")
    ALL3(7, "full-adder ripple, a and x in one bank")
    ALL3(8, "full-adder ripple, all three in different banks")
    ALL3(9, "mux, shared select, sources in three banks")
    ALL3(10, "mux, shared select, all in one bank")
    return 0;
}
