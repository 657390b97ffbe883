// h2d_rate.hip -- host->device copy rates that bound the host-pointer entry points:
// pageable hipMemcpy, hipHostRegister cost, copy from registered memory, unregister.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d tools/ubench/h2d_rate.hip && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t sizes[] = {(size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30};
    void *d = nullptr;
    if (hipMalloc(&d, sizes[2]) != hipSuccess) return 1;
    for (size_t bytes : sizes) {
        char *h = (char *)aligned_alloc(4096, bytes);
        memset(h, 1, bytes);
        hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        double t = now();
        hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        double t_page = now() - t;
        t = now();
        hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
        double t_reg = now() - t;
        t = now();
        hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        double t_pin = now() - t;
        t = now();
        hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        double t_pin_d2h = now() - t;
        t = now();
        hipHostUnregister(h);
        double t_unreg = now() - t;
        printf("%5zu MB: pageable H2D %.1f GB/s | register %.2f ms (%s) = %.1f GB/s | pinned H2D %.1f GB/s, D2H %.1f GB/s | unregister %.2f ms\n",
               bytes >> 20, bytes / t_page / 1e9, t_reg * 1e3, hipGetErrorString(e), bytes / t_reg / 1e9, bytes / t_pin / 1e9,
               bytes / t_pin_d2h / 1e9, t_unreg * 1e3);
        free(h);
    }
    return 0;
}
