// Streaming rate of the LLR conversion kernels (csrc/llr_convert.hip) with its tuning macros overridden:
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -DLLRC_NT_STORE=0 -o /tmp/llr_stream tools/ubench/llr_stream.hip && /tmp/llr_stream
#include "../../labrador_ldpc_amd/csrc/llr_convert.hip"
#include <cstdio>
template <class T> void run(const char *name, size_t bytes)
{
    uint8_t *bits; T *llrs;
    (void)hipMalloc(&bits, bytes); (void)hipMalloc(&llrs, bytes * 8 * sizeof(T));
    (void)hipMemset(bits, 0x5A, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms[2];
    for (int dir = 0; dir < 2; ++dir) {
        for (int r = 0; r < 3; ++r) dir ? (void)ldpc::launch_llrs_to_hard<T>(llrs, bits, bytes, nullptr) : (void)ldpc::launch_hard_to_llrs<T>(bits, llrs, bytes, nullptr);
        (void)hipEventRecord(a);
        for (int r = 0; r < 10; ++r) dir ? (void)ldpc::launch_llrs_to_hard<T>(llrs, bits, bytes, nullptr) : (void)ldpc::launch_hard_to_llrs<T>(bits, llrs, bytes, nullptr);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms[dir], a, b); ms[dir] /= 10;
    }
    const double total = (double)bytes * (1 + 8 * sizeof(T));
    printf("%-4s %6.2f GB: hard_to_llrs %.3f ms %5.0f GB/s | llrs_to_hard %.3f ms %5.0f GB/s\n", name, total / 1e9, ms[0], total / ms[0] / 1e6, ms[1], total / ms[1] / 1e6);
    (void)hipFree(bits); (void)hipFree(llrs);
}
int main()
{
    printf("unroll %d nt_store %d nt_load %d\n", LLRC_UNROLL, LLRC_NT_STORE, LLRC_NT_LOAD);
    run<int8_t>("i8", (size_t)1 << 30);
    run<float>("f32", (size_t)1 << 28);
    run<double>("f64", (size_t)1 << 27);
    return 0;
}
