// HBM write ceiling for two store shapes: 16 bytes per lane (1 KB per wave-instruction) and 4 bytes per lane (256 B).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_ __attribute__((ext_vector_type(4)));
template <int NT> __global__ void __launch_bounds__(256) fill16(int4_ *p, size_t n, int v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { int4_ x = {v, v + 1, v + 2, v + 3}; if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x; }
}
template <int NT> __global__ void __launch_bounds__(256) fill4(int *p, size_t n, int v)
{
    const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64, l = threadIdx.x & 63;      // a wave fills 1 KB with four 256-B stores
#pragma unroll
    for (int j = 0; j < 4; ++j) { const size_t i = w * 256 + j * 64 + l; if (i < n) { if (NT) __builtin_nontemporal_store(v + j, p + i); else p[i] = v + j; } }
}
int main()
{
    const size_t bytes = (size_t)8 << 30;
    void *d; (void)hipMalloc(&d, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto time = [&](auto launch, const char *name) {
        for (int r = 0; r < 2; ++r) launch();
        (void)hipEventRecord(a);
        for (int r = 0; r < 5; ++r) launch();
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-28s %.3f ms %5.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    const size_t n16 = bytes / 16, n4 = bytes / 4;
    time([&] { fill16<0><<<(unsigned)(n16 / 256), 256>>>((int4_ *)d, n16, 1); }, "16 B per lane");
    time([&] { fill16<1><<<(unsigned)(n16 / 256), 256>>>((int4_ *)d, n16, 1); }, "16 B per lane, nontemporal");
    time([&] { fill4<0><<<(unsigned)(n16 / 256), 256>>>((int *)d, n4, 1); }, "4 x 4 B per lane");
    time([&] { fill4<1><<<(unsigned)(n16 / 256), 256>>>((int *)d, n4, 1); }, "4 x 4 B per lane, nontemporal");
    time([&] { (void)hipMemsetAsync(d, 0, bytes, nullptr); }, "hipMemsetAsync");
    return 0;
}
