// decode_bf.hpp -- batched bit-flipping decoder (see decode_bf.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ldpc {
// input [batch][n/8] -> output [batch][(n+p)/8], iters [batch], success [batch]; device pointers.
hipError_t launch_decode_bf(int code, const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success,
                            size_t batch, uint32_t maxiters, hipStream_t stream);
}
