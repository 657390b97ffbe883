// decode_ms_i32.hip -- i32 instantiations of the min-sum kernel (decode_ms::<i32>,
// /root/reference/src/decoder.rs:60-68, :347-475; the Rust generic accepts i32 although the reference's
// C API does not export it).  Integer arithmetic throughout (Ops<int32_t>, decode_ms_kernel.hpp).
#include "decode_ms_launch.hpp"

namespace ldpc {

template <>
hipError_t launch_decode_ms<int32_t>(int code, int variant, const int32_t *llrs, uint8_t *output,
                                     uint32_t *iters, uint8_t *success, size_t batch,
                                     uint32_t maxiters, hipStream_t stream)
{
    LDPC_SPLIT_VARIANT();
    if (variant == VARIANT_PAIR || (variant == 0 && code == TM8192)) {
        if (code == TM8192) return launch_pair<TM8192, int32_t>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        return hipErrorInvalidConfiguration;
    }
    switch (code) {
        LDPC_CASE(TC128,  int32_t, 1)
        LDPC_CASE(TC256,  int32_t, 1)
        LDPC_CASE(TC512,  int32_t, 1)
        LDPC_CASE(TM1280, int32_t, 1)
        LDPC_CASE(TM1536, int32_t, 1)
        LDPC_CASE(TM2048, int32_t, 1)
        LDPC_CASE(TM5120, int32_t, 1)
        LDPC_CASE(TM6144, int32_t, 1)
        LDPC_CASE(TM8192, int32_t, 2)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ldpc
