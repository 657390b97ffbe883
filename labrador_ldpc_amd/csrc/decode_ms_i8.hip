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
                                      uint32_t maxiters, hipStream_t stream, int refill);
// the launch queue's word for the bit-sliced refill kernels (decode_ms_bs.hip does not see decode_ms_launch.hpp)
namespace bs { uint32_t *bs_queue_word(hipStream_t stream) { return claim_counter(stream); } }
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

// ONE predicate for the launcher and for the introspection entry (round 5 advice: the name used to be derived separately
// and could name a kernel the launcher refuses): which kernel family serves (code, kernel part of `variant`, flags, batch) on
// buffers of this alignment -- NONE = the launcher returns hipErrorInvalidConfiguration (EUNSUPPORTED at the C ABI).
enum class I8Kernel { NONE, BITSLICED, BITSLICED_SPLIT, PAIR, PIPE };
#define LDPC_BUILT_CASE(CODE, T, DEF, ...) case CODE: { constexpr int alts[] = {DEF, ##__VA_ARGS__}; for (int a : alts) if (a == variant) return true; return false; }
static bool pipe_variant_built(int code, int variant)          // an explicit indices-per-thread value of the table above
{
    switch (code) {
        LDPC_TABLE(LDPC_BUILT_CASE)
        default: return false;
    }
}
static I8Kernel pick_i8_kernel(int code, int variant, unsigned lflags, size_t batch, bool aligned)
{
    if (!valid_code(code) || variant < 0) return I8Kernel::NONE;
    const bool tm = code >= TM1280 && code <= TM8192;
    const I8Kernel bs = (code == TM1280 || code == TM5120) ? I8Kernel::BITSLICED_SPLIT : I8Kernel::BITSLICED;
    if (variant == VARIANT_BITSLICE) return tm && aligned ? bs : I8Kernel::NONE;
    if (variant == 0 && lflags == 0 && batch >= bitslice_min_batch(code) && aligned) return bs;
    if (variant == VARIANT_PAIR) return code == TM8192 ? I8Kernel::PAIR : I8Kernel::NONE;
    if (variant == 0) return code == TM8192 ? I8Kernel::PAIR : I8Kernel::PIPE;
    return pipe_variant_built(code, variant) ? I8Kernel::PIPE : I8Kernel::NONE;
}

// Slot refill (TM1536, TM1280: a finished codeword's lanes take the next frame at once): by name -- `variant` 64 without STATIC -- at
// any batch size; in the default dispatch from 4 frames per resident slot up (2 048 waves x 8, 1 024 wave pairs x 16 = 16 384 slots on
// 256 CUs): with fewer frames per slot there is little to refill with and the per-slot events cost more than the lockstep loses
// (tools/bs_crossover.py, profiles/r06_kbench/bs_crossover_refill.txt: TM1536 break-even at 65 536 frames, TM1280 between 32 768 and 65 536).
constexpr size_t REFILL_MIN_BATCH = 65536;
static bool i8_refills(int code, int variant, unsigned lflags, size_t batch)
{
    if ((code != TM1536 && code != TM1280) || (lflags & LF_STATIC)) return false;
    return variant == VARIANT_BITSLICE || batch >= REFILL_MIN_BATCH;
}

// the kernel labrador_ldpc_decode_ms_batch_i8 launches for a 4-byte-aligned batch of this size ("" = this build has none for the
// request: the call returns LABRADOR_LDPC_HIP_EUNSUPPORTED).  Buffers that are not 4-byte aligned never take the bit-sliced kernels.
const char *decode_ms_i8_kernel_name(int code, int variant, size_t batch)
{
    if (!valid_code(code) || variant < 0) return "";
    LDPC_SPLIT_VARIANT();
    switch (pick_i8_kernel(code, variant, lflags, batch, true)) {
        // (the slot-refill kernels: i8_refills -- and on streams that have a queue word: all but a stream under capture)
        case I8Kernel::BITSLICED:       return i8_refills(code, variant, lflags, batch) ? "decode_ms_bs_refill_kernel" : "decode_ms_bs_kernel";
        case I8Kernel::BITSLICED_SPLIT: return i8_refills(code, variant, lflags, batch) ? "decode_ms_bs_split_refill_kernel" : "decode_ms_bs_split_kernel";
        case I8Kernel::PAIR:            return "decode_ms_pair_kernel";
        case I8Kernel::PIPE:            return "decode_ms_kernel";
        default:                        return "";
    }
}

template <>
hipError_t launch_decode_ms<int8_t>(int code, int variant, const int8_t *llrs, uint8_t *output,
                                    uint32_t *iters, uint8_t *success, size_t batch,
                                    uint32_t maxiters, hipStream_t stream)
{
    LDPC_SPLIT_VARIANT();
    switch (pick_i8_kernel(code, variant, lflags, batch, bitslice_aligned(llrs, output))) {
        case I8Kernel::BITSLICED:
        case I8Kernel::BITSLICED_SPLIT:
            return launch_decode_ms_bitsliced(code, llrs, output, iters, success, batch, maxiters, stream, i8_refills(code, variant, lflags, batch) ? 1 : 0);
        case I8Kernel::PAIR:           // TM8192: pair-ownership kernel by default (decode_ms_pair.hpp), `variant` 2 / 4 = the (t, t + M/2) kernel
            return launch_pair<TM8192, int8_t>(llrs, output, iters, success, batch, maxiters, stream, lflags);
        case I8Kernel::PIPE:
            break;
        default:
            return valid_code(code) ? hipErrorInvalidConfiguration : hipErrorInvalidValue;
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
