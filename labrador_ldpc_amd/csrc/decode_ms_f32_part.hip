// decode_ms_f32_part.hip -- the heavy f32 instantiations of the min-sum kernel, compiled as three objects of their own
// (Makefile: -DF32_PART=1/2/3) so that the build stays parallel: decode_ms_f32.hip declares them `extern template`.
// (decode_ms::<f32>, /root/reference/src/decoder.rs:69-77, :347-475)
#include "decode_ms_launch.hpp"

namespace ldpc {

#ifndef F32_PART
#error "compile with -DF32_PART=1, 2 or 3"
#endif
#define LDPC_F32_SIG (const float *, uint8_t *, uint32_t *, uint8_t *, size_t, uint32_t, hipStream_t, unsigned)

#if F32_PART == 1
template hipError_t launch_pair<TM8192, float> LDPC_F32_SIG;          // the metric's kernel, both clamp forms
#elif F32_PART == 2
template hipError_t launch_pair<TM2048, float> LDPC_F32_SIG;
template hipError_t launch_one<TM8192, float, 2> LDPC_F32_SIG;
template hipError_t launch_one<TM8192, float, 4> LDPC_F32_SIG;
#else
template hipError_t launch_one<TM5120, float, 1> LDPC_F32_SIG;        // one-pass kernel and the two NaN passes (two_pass_nan())
template hipError_t launch_one<TM6144, float, 1> LDPC_F32_SIG;
template hipError_t launch_one<TM6144, float, 2> LDPC_F32_SIG;
#endif

}  // namespace ldpc
