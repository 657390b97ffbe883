// decode_bf.hip -- batched hard-decision decoder: erasure pre-pass + Gallager bit flipping.
//
// Replaces LDPCCode::decode_bf (/root/reference/src/decoder.rs:243-301) and its private
// erasure pre-pass decode_erasures (:144-223) for batches of frames; results (output bytes,
// iteration count, success) equal the reference's per frame.  One workgroup per codeword, the
// hard bits, check parities and per-variable violation counts live in LDS; a thread walks the
// sub-matrix blocks for its indices i (the same compile-time block lists as the min-sum kernel),
// so every LDS access of a wave is unit-stride with at most one wrap.
//
// Reference behaviour that is reproduced deliberately:
//  * the erasure pass counts every still-erased variable into `bits_fixed` on every iteration
//    (decoder.rs:205-213), so it always stops after its FIRST iteration with (true, 0): a
//    punctured bit becomes 1 iff the checks that have it as their only erased variable vote
//    so by majority (decoder.rs:192-210), else it stays 0;
//  * a check's erasure count is the number of its edges into punctured columns, a block-row
//    constant here (all punctured bits start erased, decoder.rs:163-164);
//  * returned iterations = bit-flipping iterations + erasure iterations (= +0), decoder.rs:289, :300.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

#include "decode_bf.hpp"
#include "decode_ms_kernel.hpp"      // static_for, pi_dev, prototype helpers

namespace ldpc {

namespace {

constexpr int BF_THREADS = 256;
constexpr int BF_BITSLICE_MIN_GROUPS = 256;

constexpr int punctured_edges_in_row(const Prototype &p, int row, int ntx)
{
    int c = 0;
    for (int b = 0; b < p.n_blocks; ++b) c += (p.blk[b].row == row && p.blk[b].col >= ntx) ? 1 : 0;
    return c;
}

template <int CODE>
__global__ void __launch_bounds__(BF_THREADS)
decode_bf_kernel(const uint8_t *__restrict__ input, uint8_t *__restrict__ output,
                 uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                 uint32_t batch, uint32_t maxiters)
{
    constexpr Prototype P = *CODES[CODE].proto;
    constexpr int M = CODES[CODE].m, N = CODES[CODE].n, NP = CODES[CODE].n + CODES[CODE].p;
    constexpr int NB = P.n_blocks, NROWS = P.n_rows, NTX = N / M, NC = CODES[CODE].n_checks();
    constexpr int LQ = ilog2(M / 4);
    constexpr int IPT = (M + BF_THREADS - 1) / BF_THREADS;       // indices per thread (strided)

    __shared__ uint8_t bits[NP];          // current hard decision per variable
    __shared__ uint32_t cnt[NP];          // violations per variable (decoder.rs:277-286) / erasure votes
    __shared__ uint32_t maxv;             // max_violations (decoder.rs:276)

    const int tid = threadIdx.x;
    auto wire = [&](auto B_, int i) LDPC_INLINE -> int {
        constexpr Block blk = P.blk[decltype(B_)::value];
        if constexpr (blk.kind == BLK_I) return blk.col * M + ((i + blk.val) & (M - 1));
        else return blk.col * M + pi_dev<blk.val, M>(i, i >> LQ);
    };

    for (uint32_t cw = blockIdx.x; cw < batch; cw += gridDim.x) {
        // ---- unpack: output[..n/8] = input (decoder.rs:251), punctured bits 0 (:167) ------------
        for (int x = tid; x < NP; x += BF_THREADS)
            bits[x] = x < N ? (input[(size_t)cw * (N / 8) + x / 8] >> (7 - x % 8)) & 1 : 0;
        for (int x = tid; x < NP; x += BF_THREADS) cnt[x] = 0;
        __syncthreads();

        // ---- erasure pre-pass (decoder.rs:144-223): one effective iteration --------------------
        if constexpr (NP > N) {
            if (maxiters > 0) {
                static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                    const int i = decltype(S_)::value * BF_THREADS + tid;
                    if (i < M) {
                        static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                            constexpr int Rw = decltype(R_)::value;
                            if constexpr (punctured_edges_in_row(P, Rw, NTX) == 1) {          // exactly one erasure (:194)
                                uint32_t par = 0;                                                // over non-erased bits (:185-188)
                                static_for<0, NB>([&](auto B_) LDPC_INLINE {
                                    constexpr Block blk = P.blk[decltype(B_)::value];
                                    if constexpr (blk.row == Rw && blk.col < NTX) par ^= bits[wire(B_, i)];
                                });
                                static_for<0, NB>([&](auto B_) LDPC_INLINE {
                                    constexpr Block blk = P.blk[decltype(B_)::value];
                                    if constexpr (blk.row == Rw && blk.col >= NTX)
                                        atomicAdd(&cnt[wire(B_, i)], par ? 1u : 0xFFFFFFFFu);   // +1 / -1 (:196-200)
                                });
                            }
                        });
                    }
                });
                __syncthreads();
                for (int x = N + tid; x < NP; x += BF_THREADS) bits[x] = (int)cnt[x] > 0 ? 1 : 0;   // :207-210
                __syncthreads();
            }
        }

        // ---- bit flipping (decoder.rs:264-298) ---------------------------------------------------
        bool ok = false;
        uint32_t iters = maxiters;                                                        // :300 (+0 erasure iterations)
        for (uint32_t it = 0; it < maxiters; ++it) {
            for (int x = tid; x < NP; x += BF_THREADS) cnt[x] = 0;                        // :266
            if (tid == 0) maxv = 0;
            __syncthreads();
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                const int i = decltype(S_)::value * BF_THREADS + tid;
                if (i < M) {
                    static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                        constexpr int Rw = decltype(R_)::value;
                        uint32_t par = 0;                                                   // :269-273
                        static_for<0, NB>([&](auto B_) LDPC_INLINE {
                            if constexpr (P.blk[decltype(B_)::value].row == Rw) par ^= bits[wire(B_, i)];
                        });
                        if (par) {                                                          // :278-281
                            static_for<0, NB>([&](auto B_) LDPC_INLINE {
                                if constexpr (P.blk[decltype(B_)::value].row == Rw) atomicAdd(&cnt[wire(B_, i)], 1u);
                            });
                        }
                    });
                }
            });
            __syncthreads();
            uint32_t m = 0;                                                                // :282-284
            for (int x = tid; x < NP; x += BF_THREADS) m = cnt[x] > m ? cnt[x] : m;
            if (m) atomicMax(&maxv, m);
            __syncthreads();
            const uint32_t mx = maxv;
            if (mx == 0) { ok = true; iters = it; break; }                                 // :288-289
            for (int x = tid; x < NP; x += BF_THREADS)                                     // :292-296
                if (cnt[x] == mx) bits[x] ^= 1;
            __syncthreads();
        }

        // ---- pack (MSB first) --------------------------------------------------------------------
        for (int j = tid; j < NP / 8; j += BF_THREADS) {
            uint32_t b = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) b |= (uint32_t)bits[8 * j + q] << (7 - q);
            output[(size_t)cw * (NP / 8) + j] = (uint8_t)b;
        }
        if (tid == 0) { iters_out[cw] = iters; success_out[cw] = ok ? 1 : 0; }
        __syncthreads();
        (void)NC;
    }
}

template <int CODE>
hipError_t launch(const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                  uint32_t maxiters, hipStream_t stream)
{
    const unsigned grid = (unsigned)(batch < 65536 ? batch : 65536);
    hipLaunchKernelGGL((decode_bf_kernel<CODE>), dim3(grid), dim3(BF_THREADS), 0, stream, input, output, iters,
                       success, (uint32_t)batch, maxiters);
    return hipGetLastError();
}

}  // namespace

// the bit-sliced kernel (decode_bf_bs.hip, decode_bf_bitslice.hpp): TM codes
hipError_t launch_decode_bf_bitsliced(int code, const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                                      uint32_t maxiters, hipStream_t stream);

hipError_t launch_decode_bf(int code, const uint8_t *input, uint8_t *output, uint32_t *iters, uint8_t *success,
                            size_t batch, uint32_t maxiters, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    // TM codes, from BF_BITSLICE_MIN_GROUPS groups of 64 / (M/32) codewords up: one wave decodes a group on its own with one register
    // per block column (decode_bf_bitslice.hpp); below that the workgroup-per-codeword kernel here has the shorter latency.
    // LABRADOR_LDPC_HIP_BF_BYTES=1 forces the byte-per-variable kernel (A/B timing, tests).
    static const bool force_bytes = [] { const char *e = std::getenv("LABRADOR_LDPC_HIP_BF_BYTES"); return e && *e && *e != '0'; }();
    if (!force_bytes && code >= TM1280 && code <= TM8192 && (uintptr_t)input % 4 == 0 && (uintptr_t)output % 4 == 0 &&
        batch >= (size_t)BF_BITSLICE_MIN_GROUPS * (size_t)(64 / (CODES[code].m / 32))) {
        // (its queue head is a stream-ordered allocation: not inside a graph capture, not on the per-thread default stream)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const bool plain = stream != hipStreamPerThread && (stream == nullptr || (hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusNone));
        if (plain) {
            const hipError_t e = launch_decode_bf_bitsliced(code, input, output, iters, success, batch, maxiters, stream);
            if (e != hipErrorOutOfMemory && e != hipErrorNotSupported) return e;
            (void)hipGetLastError();
        }
    }
    switch (code) {
        case TC128:  return launch<TC128>(input, output, iters, success, batch, maxiters, stream);
        case TC256:  return launch<TC256>(input, output, iters, success, batch, maxiters, stream);
        case TC512:  return launch<TC512>(input, output, iters, success, batch, maxiters, stream);
        case TM1280: return launch<TM1280>(input, output, iters, success, batch, maxiters, stream);
        case TM1536: return launch<TM1536>(input, output, iters, success, batch, maxiters, stream);
        case TM2048: return launch<TM2048>(input, output, iters, success, batch, maxiters, stream);
        case TM5120: return launch<TM5120>(input, output, iters, success, batch, maxiters, stream);
        case TM6144: return launch<TM6144>(input, output, iters, success, batch, maxiters, stream);
        case TM8192: return launch<TM8192>(input, output, iters, success, batch, maxiters, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ldpc
