// decode_ms_f64_reg.hip -- f64 instantiations of the register-resident min-sum kernel
// (decode_ms::<f64>, /root/reference/src/decoder.rs:78-86, :347-475) for the codes whose f64 exchange
// arrays fit the 160 KB of LDS; TM8192 (176 KB with an array of marginals) runs the in-place variant
// (LEAN == 2, 152 KB).  The workspace kernel of decode_ms_f64.hip remains as variant 100.
#include "decode_ms_launch.hpp"

namespace ldpc {

// The instantiations are compiled as three objects (Makefile: -DF64_PART=0/1/2) to keep the build parallel.
#ifndef F64_PART
#error "compile with -DF64_PART=0, 1 or 2"
#endif
#define F64_CAT2(a, b) a##b
#define F64_CAT(a, b) F64_CAT2(a, b)

#if F64_PART == 0
hipError_t launch_decode_ms_f64_reg_1(int, int, int, const double *, uint8_t *, uint32_t *, uint8_t *, size_t, uint32_t, hipStream_t);
hipError_t launch_decode_ms_f64_reg_2(int, int, int, const double *, uint8_t *, uint32_t *, uint8_t *, size_t, uint32_t, hipStream_t);
#endif

// ipt / lean select the instantiation; hipErrorInvalidConfiguration if it was not built.
#if F64_PART == 0
hipError_t launch_decode_ms_f64_reg(
#else
hipError_t F64_CAT(launch_decode_ms_f64_reg_, F64_PART)(
#endif
int code, int ipt, int lean, const double *llrs, uint8_t *output, uint32_t *iters,
                                    uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
#define F64_CFG(CODE, IPT, LEAN)                                                                                     \
    if (code == CODE && ipt == IPT && lean == LEAN)                                                                  \
        return launch_cfg<CODE, double, IPT, LEAN>(llrs, output, iters, success, batch, maxiters, stream, 0u);
#if F64_PART == 0
    F64_CFG(TC128, 1, 0)  F64_CFG(TC128, 1, 1)
    F64_CFG(TC256, 1, 0)  F64_CFG(TC256, 1, 1)
    F64_CFG(TC512, 1, 0)  F64_CFG(TC512, 1, 1)
    if (code == TM1280 || code == TM1536 || code == TM2048)
        return launch_decode_ms_f64_reg_1(code, ipt, lean, llrs, output, iters, success, batch, maxiters, stream);
    return launch_decode_ms_f64_reg_2(code, ipt, lean, llrs, output, iters, success, batch, maxiters, stream);
#elif F64_PART == 1
    F64_CFG(TM1280, 1, 0) F64_CFG(TM1280, 1, 1) F64_CFG(TM1280, 1, 2)
    F64_CFG(TM1536, 1, 0) F64_CFG(TM1536, 1, 1)
    F64_CFG(TM2048, 1, 0) F64_CFG(TM2048, 1, 1) F64_CFG(TM2048, 1, 2)
    return hipErrorInvalidConfiguration;
#else
    F64_CFG(TM5120, 1, 1)  F64_CFG(TM5120, 2, 1) F64_CFG(TM5120, 1, 2)
    F64_CFG(TM6144, 1, 1)  F64_CFG(TM6144, 2, 1) F64_CFG(TM6144, 2, 0) F64_CFG(TM6144, 1, 2)
    F64_CFG(TM8192, 2, 2)  F64_CFG(TM8192, 4, 2)
    return hipErrorInvalidConfiguration;
#endif
#undef F64_CFG
}

}  // namespace ldpc
