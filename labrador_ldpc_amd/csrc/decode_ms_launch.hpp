// decode_ms_launch.hpp -- host-side dispatch from (code, variant) to a kernel instantiation.
//
// Each LLR type has its own translation unit (decode_ms_f32.hip, ...) so the instantiations
// compile in parallel; they all expand LDPC_DEFINE_LAUNCHER below.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "decode_ms_kernel.hpp"

namespace ldpc {

// Launch the decoder for `batch` frames on `stream`.  `variant` = 0 picks the tuned default
// IPT (indices per thread) for the code; a positive value requests that IPT explicitly and
// yields hipErrorInvalidConfiguration if it was not instantiated.
template <class T>
hipError_t launch_decode_ms(int code, int variant, const T *llrs, uint8_t *output, uint32_t *iters,
                            uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream);

template <int CODE, class T, int IPT>
hipError_t launch_one(const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                      size_t batch, uint32_t maxiters, hipStream_t stream)
{
    using GEO = Geometry<CODE, T, IPT>;
    if (batch == 0) return hipSuccess;
    const size_t groups = (batch + GEO::G - 1) / GEO::G;
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((decode_ms_kernel<CODE, T, IPT>), dim3((unsigned)groups), dim3(GEO::WG), 0, stream,
                       llrs, output, iters, success, (uint32_t)batch, maxiters);
    return hipGetLastError();
}

// one `case` of the dispatch switch: default IPT plus optional alternatives
#define LDPC_CASE(CODE, T, DEF, ...)                                                             \
    case CODE: {                                                                                 \
        constexpr int alts[] = {DEF, ##__VA_ARGS__};                                             \
        return dispatch_ipt<CODE, T, DEF, ##__VA_ARGS__>(variant == 0 ? alts[0] : variant, llrs, \
                                                         output, iters, success, batch, maxiters, \
                                                         stream);                                \
    }

template <int CODE, class T, int... IPTS>
hipError_t dispatch_ipt(int ipt, const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                        size_t batch, uint32_t maxiters, hipStream_t stream)
{
    hipError_t r = hipErrorInvalidConfiguration;
    (void)((ipt == IPTS ? (r = launch_one<CODE, T, IPTS>(llrs, output, iters, success, batch, maxiters, stream), true)
                        : false) || ...);
    return r;
}

}  // namespace ldpc
