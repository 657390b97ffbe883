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
// bitslice_min_batch() frames up -- one wave decodes a group of 64 / (M/32) codewords on its own, so it needs ~2048 groups in flight
// to fill the chip and takes ~3x as long per codeword as a whole workgroup of the f32-pipe kernels: small batches are
// faster on those (and a single frame's latency is theirs).  `variant` 1 / 2 / 32 still name the f32-pipe kernels explicitly.
hipError_t launch_decode_ms_bitsliced(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream);
hipError_t launch_decode_ms_bitsliced_split(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                            uint32_t maxiters, hipStream_t stream);
constexpr int VARIANT_BITSLICE = 64;
// The rate-4/5 codes (TM1280, TM5120) have two bit-sliced kernels: a codeword group shared by the two waves of a workgroup
// (decode_ms_bitslice_split.hpp) -- what `variant` 64 and the default dispatch mean for them -- and, `variant` 128, round 4's first form:
// one wave per group with its LLR planes in a stream-ordered global workspace (kept for the A/B: profiles/r04_kbench/split_rate.txt).
constexpr int VARIANT_BITSLICE_ONE_WAVE = 128;
static hipError_t launch_bitsliced_default_form(int code, const int8_t *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                                uint32_t maxiters, hipStream_t stream)
{
    if (code == TM1280 || code == TM5120) return launch_decode_ms_bitsliced_split(code, llrs, output, iters, success, batch, maxiters, stream);
    return launch_decode_ms_bitsliced(code, llrs, output, iters, success, batch, maxiters, stream);
}
// groups of 64 / (M/32) codewords from which the bit-sliced kernel is faster per call (tools/bs_crossover.py,
// profiles/r04_kbench/bs_crossover.txt: TM8192 and TM6144 cross at 1024 groups, TM2048 at 2048, TM1536 at 8192; the rate-4/5 codes'
// two-wave kernel -- twice the waves per group -- crosses at 768 (TM5120) and 2048 (TM1280) groups: profiles/r04_kbench/split_rate.txt)
constexpr size_t bitslice_min_batch(int code)
{
    constexpr size_t groups[NUM_CODES] = {0, 0, 0, 2048, 8192, 2048, 768, 1024, 1024};
    return code >= TM1280 && code <= TM8192 ? groups[code] * (size_t)(64 / (CODES[code].m / 32)) : ~(size_t)0;
}
static bool bitslice_default(int code, const int8_t *llrs, size_t batch, hipStream_t stream)
{
    (void)stream;                                  // (no kernel of the default form allocates anything: graph capture is fine)
    return batch >= bitslice_min_batch(code) && (uintptr_t)llrs % 4 == 0;
}

template <>
hipError_t launch_decode_ms<int8_t>(int code, int variant, const int8_t *llrs, uint8_t *output,
                                    uint32_t *iters, uint8_t *success, size_t batch,
                                    uint32_t maxiters, hipStream_t stream)
{
    LDPC_SPLIT_VARIANT();
    if (variant == VARIANT_BITSLICE) {
        if ((uintptr_t)llrs % 4) return hipErrorInvalidConfiguration;        // (its loads are dwords)
        return launch_bitsliced_default_form(code, llrs, output, iters, success, batch, maxiters, stream);
    }
    if (variant == VARIANT_BITSLICE_ONE_WAVE) {
        if ((uintptr_t)llrs % 4 || (code != TM1280 && code != TM5120)) return hipErrorInvalidConfiguration;
        return launch_decode_ms_bitsliced(code, llrs, output, iters, success, batch, maxiters, stream);
    }
    if (variant == 0 && lflags == 0 && bitslice_default(code, llrs, batch, stream))
        return launch_bitsliced_default_form(code, llrs, output, iters, success, batch, maxiters, stream);
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
