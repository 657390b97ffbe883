// decode_ms_f32.hip -- f32 instantiations of the min-sum kernel (decode_ms::<f32>,
// /root/reference/src/decoder.rs:69-77, :347-475; C entry capi/src/lib.rs:113-119).
#include "decode_ms_launch.hpp"

namespace ldpc {

// instantiated in decode_ms_f32_part.hip (three more objects: this one alone took 3 min 40 s of a 4 min build)
#define LDPC_F32_SIG (const float *, uint8_t *, uint32_t *, uint8_t *, size_t, uint32_t, hipStream_t, unsigned)
extern template hipError_t launch_pair<TM8192, float> LDPC_F32_SIG;
extern template hipError_t launch_pair<TM2048, float> LDPC_F32_SIG;
extern template hipError_t launch_one<TM8192, float, 2> LDPC_F32_SIG;
extern template hipError_t launch_one<TM8192, float, 4> LDPC_F32_SIG;
extern template hipError_t launch_one<TM5120, float, 1> LDPC_F32_SIG;
extern template hipError_t launch_one<TM6144, float, 1> LDPC_F32_SIG;
extern template hipError_t launch_one<TM6144, float, 2> LDPC_F32_SIG;

// code -> default and alternative indices per thread (one table for the dispatch and for decode_ms_reads_llrs_once)
#define LDPC_TABLE(X) \
    X(TC128,  float, 1) \
    X(TC256,  float, 1) \
    X(TC512,  float, 1) \
    X(TM1280, float, 1) \
    X(TM1536, float, 1, 2) \
    X(TM2048, float, 1, 2) \
    X(TM5120, float, 1) \
    X(TM6144, float, 1, 2) \
    X(TM8192, float, 2, 4)

template <>
hipError_t launch_decode_ms<float>(int code, int variant, const float *llrs, uint8_t *output,
                                   uint32_t *iters, uint8_t *success, size_t batch,
                                   uint32_t maxiters, hipStream_t stream)
{
    LDPC_SPLIT_VARIANT();
    // TM8192 runs the pair-ownership kernel by default (6.7 vs 6.35 M codewords/s); `variant` 2 / 4 = the
    // (t, t + M/2) kernel with that many indices per thread, 32 = the pair kernel explicitly (TM8192, TM2048; for TM6144 the compiler's control-flow structurizer turns
    // its four quarter bodies into EXEC-masked loops -- 100x slower, so it is not built)
    if (variant == VARIANT_PAIR || (variant == 0 && code == TM8192)) {
        if (code == TM8192) return launch_pair<TM8192, float>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        if (code == TM2048) return launch_pair<TM2048, float>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        return hipErrorInvalidConfiguration;
    }
    switch (code) {
        LDPC_TABLE(LDPC_CASE)
        default: return hipErrorInvalidValue;
    }
}

template <>
bool decode_ms_reads_llrs_once<float>(int code, int variant)
{
    if (variant != 0) return false;
    if (code == TM8192) return true;             // the pair kernel holds its LLRs in registers
    switch (code) {
        LDPC_TABLE(LDPC_ONCE_CASE)
        default: return false;
    }
}

}  // namespace ldpc
