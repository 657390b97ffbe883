// decode_ms_i8.hip -- i8 instantiations of the min-sum kernel (decode_ms::<i8>,
// /root/reference/src/decoder.rs:42-50, :347-475; C entry capi/src/lib.rs:97-103).
#include "decode_ms_launch.hpp"

namespace ldpc {

// code -> default and alternative indices per thread (one table for the dispatch and for decode_ms_reads_llrs_once)
#define LDPC_TABLE(X) \
    X(TC128,  int8_t, 1) \
    X(TC256,  int8_t, 1) \
    X(TC512,  int8_t, 1) \
    X(TM1280, int8_t, 1) \
    X(TM1536, int8_t, 1, 2) \
    X(TM2048, int8_t, 1) \
    X(TM5120, int8_t, 1) \
    X(TM6144, int8_t, 1, 2) \
    X(TM8192, int8_t, 2)

// The bit-sliced kernel (decode_ms_bs.hip, decode_ms_bitslice.hpp): `variant` 64, and the DEFAULT for the TM codes from
// bitslice_min_batch() frames up -- one wave (rate 4/5: two) decodes a group of 64 / (M/32) codewords on its own, so it needs a few
// thousand groups in flight to fill the chip and takes ~3x as long per codeword as a whole workgroup of the f32-pipe kernels: small
// batches are faster on those (and a single frame's latency is theirs).  `variant` 1 / 2 / 32 still name the f32-pipe kernels explicitly.
hipError_t launch_decode_ms_bitsliced(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream);
constexpr int VARIANT_BITSLICE = 64;
// groups of 64 / (M/32) codewords from which the bit-sliced kernel is faster per call (tools/bs_crossover.py,
// profiles/r05_kbench/bs_crossover.txt: TM8192 and TM1280 cross at 1024 groups, TM6144 and TM5120 at 768, TM2048 and TM1536 at 2048)
constexpr size_t bitslice_min_batch(int code)
{
    constexpr size_t groups[NUM_CODES] = {0, 0, 0, 1024, 2048, 2048, 768, 768, 1024};
    return code >= TM1280 && code <= TM8192 ? groups[code] * (size_t)(64 / (CODES[code].m / 32)) : ~(size_t)0;
}
// its loads and stores are dwords: both buffers 4-byte aligned
static bool bitslice_aligned(const int8_t *llrs, const uint8_t *output) { return (uintptr_t)llrs % 4 == 0 && (uintptr_t)output % 4 == 0; }

// which kernel the default dispatch picks for an aligned batch of this size (introspection: labrador_ldpc_hip_decode_ms_i8_kernel)
const char *decode_ms_i8_kernel_name(int code, int variant, size_t batch)
{
    if (!valid_code(code) || variant < 0) return "";
    const int flags = variant & VARIANT_FLAGS;
    variant &= ~VARIANT_FLAGS;
    const bool tm = code >= TM1280;
    if ((variant == VARIANT_BITSLICE && tm) || (variant == 0 && flags == 0 && batch >= bitslice_min_batch(code)))
        return (code == TM1280 || code == TM5120) ? "decode_ms_bs_split_kernel" : "decode_ms_bs_kernel";
    if (variant == VARIANT_PAIR || (variant == 0 && code == TM8192)) return "decode_ms_pair_kernel";
    return "decode_ms_kernel";
}

template <>
hipError_t launch_decode_ms<int8_t>(int code, int variant, const int8_t *llrs, uint8_t *output,
                                    uint32_t *iters, uint8_t *success, size_t batch,
                                    uint32_t maxiters, hipStream_t stream)
{
    LDPC_SPLIT_VARIANT();
    if (variant == VARIANT_BITSLICE) {
        if (!bitslice_aligned(llrs, output)) return hipErrorInvalidConfiguration;
        return launch_decode_ms_bitsliced(code, llrs, output, iters, success, batch, maxiters, stream);
    }
    if (variant == 0 && lflags == 0 && batch >= bitslice_min_batch(code) && bitslice_aligned(llrs, output))
        return launch_decode_ms_bitsliced(code, llrs, output, iters, success, batch, maxiters, stream);
    // TM8192: pair-ownership kernel by default (decode_ms_pair.hpp), `variant` 2 / 4 = the (t, t + M/2) kernel
    if (variant == VARIANT_PAIR || (variant == 0 && code == TM8192)) {
        if (code == TM8192) return launch_pair<TM8192, int8_t>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        return hipErrorInvalidConfiguration;
    }
    switch (code) {
        LDPC_TABLE(LDPC_CASE)
        default: return hipErrorInvalidValue;
    }
}

template <>
bool decode_ms_reads_llrs_once<int8_t>(int code, int variant)
{
    if (variant != 0) return false;
    if (code == TM8192) return true;             // the pair kernel holds its LLRs in registers
    switch (code) {
        LDPC_TABLE(LDPC_ONCE_CASE)
        default: return false;
    }
}

}  // namespace ldpc
