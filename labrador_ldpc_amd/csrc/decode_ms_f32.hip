// decode_ms_f32.hip -- f32 instantiations of the min-sum kernel (decode_ms::<f32>,
// /root/reference/src/decoder.rs:69-77, :347-475; C entry capi/src/lib.rs:113-119).
#include "decode_ms_launch.hpp"

namespace ldpc {

template <>
hipError_t launch_decode_ms<float>(int code, int variant, const float *llrs, uint8_t *output,
                                   uint32_t *iters, uint8_t *success, size_t batch,
                                   uint32_t maxiters, hipStream_t stream)
{
    switch (code) {
        LDPC_CASE(TC128,  float, 1)
        LDPC_CASE(TC256,  float, 1)
        LDPC_CASE(TC512,  float, 1)
        LDPC_CASE(TM1280, float, 1)
        LDPC_CASE(TM1536, float, 1, 2)
        LDPC_CASE(TM2048, float, 1, 2)
        LDPC_CASE(TM5120, float, 1)
        LDPC_CASE(TM6144, float, 1, 2)
        LDPC_CASE(TM8192, float, 2, 4)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ldpc
