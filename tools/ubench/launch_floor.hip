// launch_floor.hip -- what a synchronous one-kernel call costs on this box, by the way the host learns of completion: the floor under
// the reference-shaped single-frame entries (capi.hip host_pipeline, direct small-call path).
//   hipcc --offload-arch=gfx950 -O2 -o build/ub/launch_floor tools/ubench/launch_floor.hip && build/ub/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(const uint32_t *in, uint32_t *out) { if (threadIdx.x == 0) out[0] = in[0] + 1; }
__global__ void flag_kernel(const uint32_t *in, uint32_t *out, uint32_t *flag, uint32_t ticket)
{
    if (threadIdx.x == 0) {
        out[0] = in[0] + 1;
        __threadfence_system();
        __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class F> static double median_us(F f, int reps = 7, int calls = 2000)
{
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        auto a = std::chrono::steady_clock::now();
        for (int i = 0; i < calls; ++i) f();
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / calls);
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2];
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, 4096, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    h[0] = 1; h[16] = 0; h[32] = 0;
    volatile uint32_t *flag = h + 32;
    uint32_t ticket = 0;
    hipEvent_t ev, ev_spin;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming));
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); CK(hipStreamSynchronize(st)); }

    std::printf("one-wave kernel on pinned mapped memory, us per synchronous call (median of 7 x 2000)\n");
    std::printf("  launch + hipStreamSynchronize            %6.2f\n", median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); (void)hipStreamSynchronize(st); }));
    std::printf("  launch + hipStreamQuery spin             %6.2f\n", median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); while (hipStreamQuery(st) == hipErrorNotReady) {} }));
    std::printf("  launch + event record + hipEventSynchronize %6.2f\n", median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); (void)hipEventRecord(ev, st); (void)hipEventSynchronize(ev); }));
    std::printf("  launch, kernel stores a flag, host spins  %6.2f\n", median_us([&] { ++ticket; hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, st, d, d + 16, d + 32, ticket); while (*flag != ticket) {} }));
    CK(hipStreamSynchronize(st));
    std::printf("  launch + hipStreamWriteValue32, host spins %6.2f\n", median_us([&] { ++ticket; hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); (void)hipStreamWriteValue32(st, d + 32, ticket, 0); while (*flag != ticket) {} }));
    CK(hipStreamSynchronize(st));
    std::printf("  launch only (no wait; queue drained every 2000) %6.2f\n", median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d, d + 16); }, 7, 2000));
    CK(hipStreamSynchronize(st));
    // the same on the legacy null stream
    std::printf("  null stream: launch + hipStreamSynchronize %6.2f\n", median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, nullptr, d, d + 16); (void)hipStreamSynchronize(nullptr); }));
    std::printf("  out %u\n", h[16]);
    return 0;
}
