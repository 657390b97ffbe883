// decode_ms_f64.hip -- f64 min-sum decoder (decode_ms::<f64>, /root/reference/src/decoder.rs:78-86,
// :347-475; C entry capi/src/lib.rs:121-127).
//
// By default every code runs the register-resident kernel of decode_ms_kernel.hpp with 64-bit
// registers and LDS elements (decode_ms_f64_reg.hip): plain for the small codes, register-lean for
// TM2048 / TM5120, "in place" (no array of marginals, v kept as sign/zero bit masks) for TM6144 and
// TM8192, whose f64 exchange arrays would not fit the LDS otherwise.  f64 LLRs are the least used
// variant of the reference's API; the kernel below is the general fallback (`variant` 100, 10-100x
// slower) that trades
// speed for generality: one workgroup per codeword, marginals in LDS, the per-edge messages u and v
// in a global-memory workspace (edge e = block * M + check index: coalesced), per-check minima in
// registers of the thread that owns the check.  Same block lists, same arithmetic order:
//   variable phase  thread x of block column c:  va = llr + sum of u over its blocks in list order
//                   (decoder.rs:382-383, :408; the check index of a block is the closed-form
//                   inverse of the block's rotation)
//   check phase     thread i of block row r:     decoder.rs:419-447 for its edges in order, then
//                   decoder.rs:391-405 (next u) from the row's (min1, min2, sign)
// Results equal the reference's bit for bit (IEEE f64 add/sub, no contraction).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "decode_ms_kernel.hpp"      // static_for, prototype helpers, phi/theta accessors
#include "decode_ms_launch.hpp"

namespace ldpc {

namespace {


// check index i (inside the block) connected to variable x of block B: inverse of block_map()
template <int K, int M>
LDPC_DEV int pi_inv_dev(int x)
{
    constexpr int LQ = ilog2(M / 4), Q = M / 4;
    constexpr int TH = theta_of(K);
    const int j = ((x >> LQ) - TH) & 3;                       // source quarter
    const int phi = j == 0 ? phi_of(K, 0, M) : (j == 1 ? phi_of(K, 1, M) : (j == 2 ? phi_of(K, 2, M) : phi_of(K, 3, M)));
    return (j << LQ) + ((x - phi) & (Q - 1));
}

template <int CODE, int F64_THREADS>
__global__ void __launch_bounds__(F64_THREADS)
decode_ms_f64_kernel(const double *__restrict__ llrs, uint8_t *__restrict__ output,
                     uint32_t *__restrict__ iters_out, uint8_t *__restrict__ success_out,
                     double *__restrict__ workspace, uint32_t batch, uint32_t maxiters)
{
    constexpr Prototype P = *CODES[CODE].proto;
    constexpr int M = CODES[CODE].m, N = CODES[CODE].n, NP = N + CODES[CODE].p;
    constexpr int NB = P.n_blocks, NROWS = P.n_rows, NCOLS = P.n_cols, NTX = N / M;
    constexpr int E = NB * M, LQ = ilog2(M / 4);
    constexpr int IPT = (M + F64_THREADS - 1) / F64_THREADS;

    __shared__ double va[NP];             // marginals (decoder.rs:377)
    __shared__ int unsat;

    const int tid = threadIdx.x;
    double *u = workspace + (size_t)blockIdx.x * 2 * E;       // decoder.rs:375
    double *v = u + E;                                        // decoder.rs:376

    auto wire = [&](auto B_, int i) LDPC_INLINE -> int {     // variable (in its column) of check i
        constexpr Block blk = P.blk[decltype(B_)::value];
        if constexpr (blk.kind == BLK_I) return (i + blk.val) & (M - 1);
        else return pi_dev<blk.val, M>(i, i >> LQ);
    };
    auto unwire = [&](auto B_, int x) LDPC_INLINE -> int {   // check (in its row) of variable x
        constexpr Block blk = P.blk[decltype(B_)::value];
        if constexpr (blk.kind == BLK_I) return (x - blk.val) & (M - 1);
        else return pi_inv_dev<blk.val, M>(x);
    };

    for (uint32_t cw = blockIdx.x; cw < batch; cw += gridDim.x) {
        for (int e = tid; e < E; e += F64_THREADS) { u[e] = 0.0; v[e] = 0.0; }   // decoder.rs:374
        for (int x = tid; x < NP; x += F64_THREADS) va[x] = 0.0;
        __syncthreads();

        bool ok = false;
        uint32_t iters = maxiters;
        for (uint32_t it = 0; it < maxiters; ++it) {
            // ---- variable phase ------------------------------------------------------------
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                const int x = decltype(S_)::value * F64_THREADS + tid;
                if (x < M) {
                    static_for<0, NCOLS>([&](auto C_) LDPC_INLINE {
                        constexpr int C = decltype(C_)::value;
                        double acc = 0.0;                                              // :383
                        if constexpr (C < NTX) acc = llrs[(size_t)cw * N + C * M + x]; // :382
                        static_for<0, NB>([&](auto B_) LDPC_INLINE {
                            constexpr int B = decltype(B_)::value;
                            if constexpr (P.blk[B].col == C) acc = acc + u[B * M + unwire(B_, x)];   // :408
                        });
                        va[C * M + x] = acc;
                    });
                }
            });
            if (tid == 0) unsat = 0;
            __syncthreads();

            // ---- check phase ------------------------------------------------------------------
            bool fail = false;
            static_for<0, IPT>([&](auto S_) LDPC_INLINE {
                const int i = decltype(S_)::value * F64_THREADS + tid;
                if (i < M) {
                    static_for<0, NROWS>([&](auto R_) LDPC_INLINE {
                        constexpr int Rw = decltype(R_)::value;
                        double m1 = DBL_MAX, m2 = DBL_MAX;                             // :414-415
                        bool sgn = false, par = false;                                 // :416-417
                        static_for<0, NB>([&](auto B_) LDPC_INLINE {
                            constexpr int B = decltype(B_)::value;
                            if constexpr (P.blk[B].row == Rw) {
                                const int e = B * M + i;
                                const double x = va[P.blk[B].col * M + wire(B_, i)];
                                const double nv = x - u[e];                            // :421
                                const double old = v[e];
                                const double nw = ((nv < 0.0) == (old < 0.0) || old == 0.0) ? nv : 0.0;   // :422-425
                                v[e] = nw;
                                const double a = __builtin_fabs(nw);
                                if (a < m1) { m2 = m1; m1 = a; } else if (a < m2) { m2 = a; }   // :430-435
                                sgn ^= nw < 0.0;                                        // :439-441
                                par ^= x < 0.0;                                         // :445-447
                            }
                        });
                        static_for<0, NB>([&](auto B_) LDPC_INLINE {                   // next iteration's :391-405
                            constexpr int B = decltype(B_)::value;
                            if constexpr (P.blk[B].row == Rw) {
                                const int e = B * M + i;
                                const double w = v[e];
                                double un = (__builtin_fabs(w) == m1) ? m2 : m1;
                                if (sgn) un = -un;
                                if (w < 0.0) un = -un;
                                u[e] = un;
                            }
                        });
                        fail |= par;
                    });
                }
            });
            if (fail) unsat = 1;
            __syncthreads();                       // u, v (global) and the vote are visible to the workgroup
            if (unsat == 0) { ok = true; iters = it; break; }                         // :453-463
            __syncthreads();                       // everyone has read the vote before it is reset
        }

        // ---- hard decision, MSB first (decoder.rs:455-461 / :467-473) ------------------------------
        for (int j = tid; j < NP / 8; j += F64_THREADS) {
            uint32_t b = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) b |= (uint32_t)(va[8 * j + q] < 0.0) << (7 - q);
            output[(size_t)cw * (NP / 8) + j] = (uint8_t)b;
        }
        if (tid == 0) { iters_out[cw] = iters; success_out[cw] = ok ? 1 : 0; }
        __syncthreads();
    }
}

template <int CODE, int F64_THREADS = 256, int GRID = 1024>
hipError_t launch_f64(const double *llrs, uint8_t *output, uint32_t *iters, uint8_t *success, size_t batch,
                      uint32_t maxiters, hipStream_t stream)
{
    constexpr size_t E = (size_t)CODES[CODE].proto->n_blocks * CODES[CODE].m;
    const unsigned grid = (unsigned)(batch < GRID ? batch : GRID);
    double *ws = nullptr;
    hipError_t e = hipMallocAsync((void **)&ws, (size_t)grid * 2 * E * sizeof(double), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_ms_f64_kernel<CODE, F64_THREADS>), dim3(grid), dim3(F64_THREADS), 0, stream, llrs, output, iters,
                       success, ws, (uint32_t)batch, maxiters);
    e = hipGetLastError();
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

}  // namespace

hipError_t launch_decode_ms_f64_reg(int code, int ipt, int lean, const double *llrs, uint8_t *output, uint32_t *iters,
                                    uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream);

// the tuned default per code, as a `variant` (re-measured on round 3's kernels, tools/f64_variants.py, M codewords/s: TM1280
// plain 18.2 / lean 25.6 / in place 21.3; TM6144 in place 3.37 / lean 4.04 / lean with two indices 3.82; TM1536 plain 19.5 =
// lean 19.4; the TC codes plain)
static constexpr int F64_TUNED[NUM_CODES] = {1, 1, 1, 17, 1, 17, 17, 17, 34};

// only the plain register kernel holds its f64 LLRs in registers (kernel_reads_llrs_once)
template <>
bool decode_ms_reads_llrs_once<double>(int code, int variant)
{
    if (variant != 0 || !valid_code(code)) return false;
    return (F64_TUNED[code] & (16 | 32)) == 0;
}

// variant: 0 = tuned default; 100 = the workspace kernel above; otherwise the register kernel with
// IPT = variant & 15, the register-lean check phase if variant & 16, in-place messages if variant & 32.
template <>
hipError_t launch_decode_ms<double>(int code, int variant, const double *llrs, uint8_t *output, uint32_t *iters,
                                    uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    if (!valid_code(code)) return hipErrorInvalidValue;
    if (variant >= 0) variant &= ~VARIANT_FLAGS;           // (the f64 kernels always draw from the launch's queue)
    if (variant == 0) variant = F64_TUNED[code];
    if (variant != 100)
        return launch_decode_ms_f64_reg(code, variant & 15, (variant & 32) ? 2 : ((variant & 16) ? 1 : 0), llrs, output, iters, success, batch,
                                        maxiters, stream);
    switch (code) {
        case TC128:  return launch_f64<TC128>(llrs, output, iters, success, batch, maxiters, stream);
        case TC256:  return launch_f64<TC256>(llrs, output, iters, success, batch, maxiters, stream);
        case TC512:  return launch_f64<TC512>(llrs, output, iters, success, batch, maxiters, stream);
        case TM1280: return launch_f64<TM1280>(llrs, output, iters, success, batch, maxiters, stream);
        case TM1536: return launch_f64<TM1536>(llrs, output, iters, success, batch, maxiters, stream);
        case TM2048: return launch_f64<TM2048>(llrs, output, iters, success, batch, maxiters, stream);
        case TM5120: return launch_f64<TM5120>(llrs, output, iters, success, batch, maxiters, stream);
        case TM6144: return launch_f64<TM6144>(llrs, output, iters, success, batch, maxiters, stream);
        case TM8192: return launch_f64<TM8192, 1024, 256>(llrs, output, iters, success, batch, maxiters, stream);   // 0.122 vs 0.081 M/s at 256 threads x 1024 workgroups
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ldpc
