// Micro-benchmark: LDS cost per wave instruction by access width (gfx950), unit-stride lanes, 16 waves per CU.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_rate tools/ubench/lds_rate.hip && /tmp/lds_rate
// Question (round 3): does a 16-bit LDS element (ds_read_u16 / ds_write_b16, or their d16_hi forms) cost fewer LDS
// cycles per wave instruction than a 32-bit one, i.e. would halving the exchange arrays' element width relieve the
// LDS pipe of the IPT = 1 kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) spin(float *out, int loops)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int t = threadIdx.x;
    unsigned a4 = (unsigned)(size_t)lds + t * 4, a2 = (unsigned)(size_t)lds + t * 2, a8 = (unsigned)(size_t)lds + (t & 511) * 8, a16 = (unsigned)(size_t)lds + (t & 255) * 16;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 p0 = {0, 0}, p1 = {0, 0};
    f4 q0 = {0, 0, 0, 0};
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) asm volatile(REP8("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4096\n ds_read_b32 %2, %4 offset:8192\n ds_read_b32 %3, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a4));
        if (KIND == 1) asm volatile(REP8("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:4096\n ds_read_u16 %2, %4 offset:8192\n ds_read_u16 %3, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a2));
        if (KIND == 2) asm volatile(REP8("ds_read_u16_d16_hi %0, %4\n ds_read_u16_d16_hi %1, %4 offset:4096\n ds_read_u16_d16_hi %2, %4 offset:8192\n ds_read_u16_d16_hi %3, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a2));
        if (KIND == 3) asm volatile(REP8("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8192\n ds_read_b64 %0, %2 offset:16384\n ds_read_b64 %1, %2 offset:24576\n") "s_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(a8));
        if (KIND == 4) asm volatile(REP8("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:4096\n ds_write_b32 %0, %3 offset:8192\n ds_write_b32 %0, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" :: "v"(a4), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        if (KIND == 5) asm volatile(REP8("ds_write_b16 %0, %1\n ds_write_b16 %0, %2 offset:4096\n ds_write_b16 %0, %3 offset:8192\n ds_write_b16 %0, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" :: "v"(a2), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        if (KIND == 6) asm volatile(REP8("ds_write_b16_d16_hi %0, %1\n ds_write_b16_d16_hi %0, %2 offset:4096\n ds_write_b16_d16_hi %0, %3 offset:8192\n ds_write_b16_d16_hi %0, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" :: "v"(a2), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        if (KIND == 7) asm volatile(REP8("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8192\n ds_write_b64 %0, %1 offset:16384\n ds_write_b64 %0, %2 offset:24576\n") "s_waitcnt lgkmcnt(0)" :: "v"(a8), "v"(p0), "v"(p1) : "memory");
        if (KIND == 8) asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:8192\n ds_read_b128 %0, %1 offset:12288\n") "s_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(a16));
        if (KIND == 9) asm volatile(REP8("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:4096\n ds_read_u8 %2, %4 offset:8192\n ds_read_u8 %3, %4 offset:12288\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a2 - t));
    }
    if (r0 + r1 + r2 + r3 + p0.x + p1.y + q0.x == 12345.f) out[0] = r0;
}
template <int KIND> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int loops = 2000, blocks = 256, threads = 1024;
    spin<KIND><<<blocks, threads>>>(d, 10);
    (void)hipEventRecord(a);
    spin<KIND><<<blocks, threads>>>(d, loops);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    // per CU: 16 waves x loops x 32 instructions
    printf("%-22s %.3f ns per wave-instruction per CU\n", name, ms * 1e6 / ((double)loops * 32 * 16));
}
int main()
{
    run<0>("ds_read_b32"); run<1>("ds_read_u16"); run<2>("ds_read_u16_d16_hi"); run<3>("ds_read_b64"); run<8>("ds_read_b128"); run<9>("ds_read_u8");
    run<4>("ds_write_b32"); run<5>("ds_write_b16"); run<6>("ds_write_b16_d16_hi"); run<7>("ds_write_b64");
    return 0;
}
