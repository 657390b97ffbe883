// channel.hpp -- device-side AWGN frame generator (see channel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

namespace ldpc {
template <class T>
hipError_t launch_awgn(const uint8_t *codewords, size_t pool, T *llrs, int n, uint64_t first_frame, size_t batch, float sigma,
                       float scale, int lim, uint64_t seed, hipStream_t stream);
// shader clock (MHz) of the current device under a full-chip VALU load of `busy_ms` milliseconds; synchronous
hipError_t shader_clock_mhz(double busy_ms, double *mhz);
}
