// llr_convert.hpp -- batched hard-bit <-> LLR conversions on the device (see llr_convert.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ldpc {
// bits [bytes] (MSB first) -> llrs [8 * bytes]: -1 for a set bit, +1 for a clear one (decoder.rs:484-493).
// Device pointers, asynchronous on `stream`; llrs 16-byte aligned, `bytes` even for int8_t (every code's n/8 is).
template <class T>
hipError_t launch_hard_to_llrs(const uint8_t *bits, T *llrs, size_t bytes, hipStream_t stream);
// llrs [8 * bytes] -> bits [bytes]: bit set where the LLR is < 0 (decoder.rs:498-509; -0.0 and NaN give 0).
template <class T>
hipError_t launch_llrs_to_hard(const T *llrs, uint8_t *bits, size_t bytes, hipStream_t stream);
}
