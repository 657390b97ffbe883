// decode_ms_i32.hip -- i32 instantiations of the min-sum kernel (decode_ms::<i32>,
// /root/reference/src/decoder.rs:60-68, :347-475; the Rust generic accepts i32 although the reference's
// C API does not export it).  Integer arithmetic throughout (Ops<int32_t>, decode_ms_kernel.hpp).
#include "decode_ms_launch.hpp"

namespace ldpc {

// code -> default and alternative indices per thread (one table for the dispatch and for decode_ms_reads_llrs_once)
#define LDPC_TABLE(X) \
    X(TC128,  int32_t, 1) \
    X(TC256,  int32_t, 1) \
    X(TC512,  int32_t, 1) \
    X(TM1280, int32_t, 1) \
    X(TM1536, int32_t, 1) \
    X(TM2048, int32_t, 1) \
    X(TM5120, int32_t, 1) \
    X(TM6144, int32_t, 1) \
    X(TM8192, int32_t, 2)

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
        LDPC_TABLE(LDPC_CASE)
        default: return hipErrorInvalidValue;
    }
}

template <>
bool decode_ms_reads_llrs_once<int32_t>(int code, int variant)
{
    if (variant != 0) return false;
    if (code == TM8192) return true;             // the pair kernel holds its LLRs in registers
    switch (code) {
        LDPC_TABLE(LDPC_ONCE_CASE)
        default: return false;
    }
}

}  // namespace ldpc
