// encode.hpp -- batched GPU encoder (see encode.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ldpc {
// data [batch][k/8] -> codewords [batch][n/8], device pointers (4-byte aligned), asynchronous on `stream`.
hipError_t launch_encode(int code, const uint8_t *data, uint8_t *codewords, size_t batch, hipStream_t stream);
}
