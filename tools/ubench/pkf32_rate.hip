// Micro-benchmark (round 5): the packed-f32 VOP3P instructions of gfx950 -- v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32 (two f32 per lane
// and instruction, operands in aligned register pairs) -- against their scalar forms, at 4 / 2 / 1 waves per SIMD.  The f32 min-sum
// kernel's two indices per thread would feed them naturally; what matters is what one such instruction costs the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkf32_rate tools/ubench/pkf32_rate.hip && /tmp/pkf32_rate
// Same harness as sdwa_rate.hip: every wave executes LOOPS x 32 groups of the listed instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(INS) asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
#define BODY1(INS) asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS) spin(float *out, int loops, float seed)
{
    f2 a = {threadIdx.x * 1e-3f + seed, 1.0f}, b = {2.0f, seed}, c = {seed, 3.0f}, d = {0.5f, 0.25f}, e = {1.0000001f, 0.9999999f}, f = {1e-7f, -1e-7f};
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) { BODY1("v_add_f32 %0, %4, %1\n v_add_f32 %1, %5, %2\n v_add_f32 %2, %4, %3\n v_add_f32 %3, %5, %0") }            // (low halves only)
        if (KIND == 1) { BODY1("v_fma_f32 %0, %4, %1, %5\n v_fma_f32 %1, %4, %2, %5\n v_fma_f32 %2, %4, %3, %5\n v_fma_f32 %3, %4, %0, %5") }
        if (KIND == 2) { BODY("v_pk_add_f32 %0, %4, %1\n v_pk_add_f32 %1, %5, %2\n v_pk_add_f32 %2, %4, %3\n v_pk_add_f32 %3, %5, %0") }
        if (KIND == 3) { BODY("v_pk_mul_f32 %0, %4, %1\n v_pk_mul_f32 %1, %4, %2\n v_pk_mul_f32 %2, %4, %3\n v_pk_mul_f32 %3, %4, %0") }
        if (KIND == 4) { BODY("v_pk_fma_f32 %0, %4, %1, %5\n v_pk_fma_f32 %1, %4, %2, %5\n v_pk_fma_f32 %2, %4, %3, %5\n v_pk_fma_f32 %3, %4, %0, %5") }
        // alternating with a full-rate scalar instruction (does the packed one only cost its own slot?)
        if (KIND == 5) {
            asm volatile(REP8("v_pk_add_f32 %0, %4, %1\n v_add_f32 %2, %5, %3\n v_pk_add_f32 %1, %4, %0\n v_add_f32 %3, %5, %2\n") REP8("v_pk_add_f32 %0, %4, %1\n v_add_f32 %2, %5, %3\n v_pk_add_f32 %1, %4, %0\n v_add_f32 %3, %5, %2\n")
                         REP8("v_pk_add_f32 %0, %4, %1\n v_add_f32 %2, %5, %3\n v_pk_add_f32 %1, %4, %0\n v_add_f32 %3, %5, %2\n") REP8("v_pk_add_f32 %0, %4, %1\n v_add_f32 %2, %5, %3\n v_pk_add_f32 %1, %4, %0\n v_add_f32 %3, %5, %2\n")
                         : "+v"(a), "+v"(b), "+v"(c.x), "+v"(d.x) : "v"(e), "v"(f.x));
        }
        // a dependent chain of packed adds (latency)
        if (KIND == 6) { BODY("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %0, %5, %0\n v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %0, %5, %0") }
        if (KIND == 7) { BODY1("v_add_f32 %0, %4, %0\n v_add_f32 %0, %5, %0\n v_add_f32 %0, %4, %0\n v_add_f32 %0, %5, %0") }
        // v_pk_mov_b32 (the packing move the compiler inserts) and the op_sel forms that read halves crosswise
        if (KIND == 8) { BODY("v_pk_mov_b32 %0, %1, %2\n v_pk_mov_b32 %1, %2, %3\n v_pk_mov_b32 %2, %3, %0\n v_pk_mov_b32 %3, %0, %1") }
        if (KIND == 9) { BODY("v_pk_add_f32 %0, %4, %1 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_f32 %1, %5, %2 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_f32 %2, %4, %3 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_f32 %3, %5, %0 op_sel:[1,0] op_sel_hi:[0,1]") }
    }
    if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.0f) out[0] = a.x;
}
template <int KIND, int THREADS = 1024> void run(const char *name, int per_group = 4)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 3000, blocks = 256;
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, 10, 1.0f);
    (void)hipEventRecord(a);
    spin<KIND, THREADS><<<blocks, THREADS>>>(d, loops, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const int waves_per_simd = THREADS / 256;
    printf("%-58s %d w/SIMD  %.3f ns per wave-instr per SIMD\n", name, waves_per_simd, ms * 1e6 / ((double)loops * 32 * per_group * waves_per_simd));
    (void)hipFree(d);
}
#define ALL3(K, NAME, G) run<K, 1024>(NAME, G); run<K, 512>(NAME, G); run<K, 256>(NAME, G);
int main()
{
    ALL3(0, "v_add_f32", 4) ALL3(1, "v_fma_f32", 4)
    ALL3(2, "v_pk_add_f32 (2 adds per lane)", 4) ALL3(3, "v_pk_mul_f32", 4) ALL3(4, "v_pk_fma_f32", 4)
    ALL3(5, "v_pk_add_f32 alternating with v_add_f32", 4)
    ALL3(6, "v_pk_add_f32, one dependent chain", 4) ALL3(7, "v_add_f32, one dependent chain", 4)
    ALL3(8, "v_pk_mov_b32", 4) ALL3(9, "v_pk_add_f32 with crosswise op_sel", 4)
    return 0;
}
