// kbench.hip -- quick kernel-only timing loop for decode_ms_kernel experiments (development tool).
// On the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-fast-math -ffp-contract=off -fno-slp-vectorize \
//                  -Ilabrador_ldpc_amd/csrc [-DKCODE=8 -DKIPT=2 -DKFRAMES=131072] -o /tmp/kbench tools/kbench.hip labrador_ldpc_amd/csrc/channel.hip && /tmp/kbench
// Frames: all-zero codeword (valid for a linear code; min-sum is symmetric) + AWGN at 2 dB.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>
#define LDPC_KBENCH 1          // the only place the kernels' experiment switches may be set (decode_ms_tuning.hpp)
#include "decode_ms_kernel.hpp"
#include "decode_ms_pair.hpp"
#include "channel.hpp"
#ifndef KCODE
#define KCODE 8
#endif
#ifndef KIPT
#define KIPT 2
#endif
#ifndef KFRAMES
#define KFRAMES 131072
#endif
#ifndef KT
#define KT float
#endif
#ifndef KNANPASS
#define KNANPASS 0            // decode_ms_body's NANPASS: 1 = the first of the two NaN passes (marks, does not handle), 2 = the second
#endif
#ifndef KLEAN
#define KLEAN false
#endif
#ifndef KPAIR
#define KPAIR 0
#endif
#ifndef KPF
#define KPF false
#endif
#ifndef KMAXIT
#define KMAXIT 25
#endif
#ifndef KEBN0
#define KEBN0 2.0
#endif
#ifndef KLIMIT
#define KLIMIT 0x1p12f        // the launcher's nocap_limit_for(25, clamp form)
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    using namespace ldpc;
    constexpr int code = KCODE;
    const CodeInfo &ci = CODES[code];
    const size_t FMAX = KFRAMES, n = ci.n, ol = ci.output_len();
    size_t F = FMAX;
    if (const char *e = getenv("KB_F")) { F = (size_t)atoll(e); if (F < 1 || F > FMAX) F = FMAX; }   // fewer frames of the same allocation
    const float sigma = (float)std::sqrt(1.0 / (2.0 * ((double)ci.k / ci.n) * std::pow(10.0, KEBN0 / 10.0)));
    uint8_t *pool; KT *llrs; uint8_t *out, *ok; uint32_t *iters;
    CK(hipMalloc(&pool, n / 8)); CK(hipMemset(pool, 0, n / 8));
    CK(hipMalloc(&llrs, FMAX * n * sizeof(KT))); CK(hipMalloc(&out, FMAX * ol)); CK(hipMalloc(&ok, FMAX)); CK(hipMalloc(&iters, FMAX * 4));
    CK(launch_awgn<KT>(pool, 1, llrs, (int)n, 0, FMAX, sigma, 8.f, 31, 0x1DBCull + code, nullptr));
    using GEO = Geometry<code, KT, KIPT>;
    unsigned groups = (unsigned)((F + GEO::G - 1) / GEO::G);
    {   // the launcher's grid (decode_ms_launch.hpp): the resident set with the queue, 16x it (where several workgroups share a CU) without
        int per_cu = 1, cus = 256;
#if KPAIR
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_pair_kernel<code, KT>, PairGeometry<code, KT>::NT, 0));
        const unsigned chunks = groups;
#else
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_ms_kernel<code, KT, KIPT, KPF, KLEAN, selfcorr_med3<code, KT>(), KNANPASS>, GEO::WG, 0));
#ifdef KSTATIC
        const unsigned K = 1, chunks = groups;
#else
        const unsigned K = claim_chunk<code, KT, KIPT>(), chunks = (groups + K - 1) / K;
#endif
#endif
        CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
        unsigned resident = (unsigned)(per_cu * cus);
#ifdef KSTATIC
        unsigned grid = resident <= 256 ? resident : resident * 16;
#else
        unsigned grid = resident;
#endif
#ifdef KGRID
        grid = KGRID;
#endif
        if (const char *e = getenv("KB_GRID")) { if (atoi(e) > 0) grid = (unsigned)atoi(e); }
        groups = grid < chunks ? grid : chunks;
        printf("occupancy %d workgroups per CU x %d CUs; grid %u\n", per_cu, cus, groups);
    }
    uint32_t *claim = nullptr;
#if !defined(KSTATIC)
    CK(hipMalloc(&claim, 4)); CK(hipMemset(claim, 0, 4));       // the launch's queue head (self-resetting)
#endif
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    int burst = 1;                                             // launches between the two events (KB_BURST): back to back, no host gap
    if (const char *e = getenv("KB_BURST")) { burst = atoi(e); if (burst < 1) burst = 1; }
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        for (int bi = 0; bi < burst; ++bi)
#if KPAIR
        hipLaunchKernelGGL((decode_ms_pair_kernel<code, KT>), dim3(groups), dim3(PairGeometry<code, KT>::NT), 0, 0, llrs, out, iters, ok, (uint32_t)F, (uint32_t)KMAXIT, KLIMIT, claim);
#else
        hipLaunchKernelGGL((decode_ms_kernel<code, KT, KIPT, KPF, KLEAN, selfcorr_med3<code, KT>(), KNANPASS>), dim3(groups), dim3(GEO::WG), 0, 0, llrs, out, iters, ok, (uint32_t)F, (uint32_t)KMAXIT, KLIMIT, claim, (uint32_t)claim_chunk<code, KT, KIPT>());
#endif
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ms /= burst;
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<uint32_t> hi(F); std::vector<uint8_t> hk(F), ho(F * ol);
    CK(hipMemcpy(hi.data(), iters, F * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hk.data(), ok, F, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ho.data(), out, F * ol, hipMemcpyDeviceToHost));
    double si = 0, sk = 0; unsigned long long h = 1469598103934665603ull;
    for (size_t f = 0; f < F; ++f) { si += hi[f]; sk += hk[f]; h = (h ^ hi[f] ^ ((unsigned long long)hk[f] << 32)) * 1099511628211ull; }
    for (size_t i = 0; i < F * ol; ++i) h = (h ^ ho[i]) * 1099511628211ull;
#ifdef LDPC_DIAG_STAMPS
    {
        std::vector<unsigned long long> st(256 * 16 * 6);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ldpc::g_stamps), st.size() * 8));
        double tot[6] = {0, 0, 0, 0, 0, 0}, perq[4][6] = {};
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) for (int k = 0; k < 6; ++k) { tot[k] += st[(b * 16 + w) * 6 + k]; perq[w / 4][k] += st[(b * 16 + w) * 6 + k]; }
        const double launches = 4.0;                       // the rep loop above: 1 warm + 3 timed launches accumulate
        const double iters_total = (si + F) * launches;    // + F: the final verdict pass
        printf("stamps (s_memtime ticks per wave and iteration, all launches): variable %.0f | wait2 %.0f | check %.0f | wait1 %.0f\n",
               tot[0] / 4096 / (iters_total / 256) , tot[1] / 4096 / (iters_total / 256), tot[2] / 4096 / (iters_total / 256), tot[3] / 4096 / (iters_total / 256));
        printf("  whole codeword loop: %.0f s_memtime ticks per codeword, of which %.0f inside the iterations (%.1f %%); shader clock %.0f MHz (s_memtime against the 100 MHz s_memrealtime)\n",
               tot[4] / 4096 / (launches * F / 256), (tot[0] + tot[1] + tot[2] + tot[3]) / 4096 / (launches * F / 256), 100.0 * (tot[0] + tot[1] + tot[2] + tot[3]) / tot[4], 100.0 * tot[4] / tot[5]);
        {
            double mx = 0, mn = 1e30, mxr = 0, mnr = 1e30;
            for (int b = 0; b < 256; ++b) {
                const double v = st[(b * 16) * 6 + 4] / launches, r = st[(b * 16) * 6 + 5] / launches;
                mx = v > mx ? v : mx; mn = v < mn ? v : mn; mxr = r > mxr ? r : mxr; mnr = r < mnr ? r : mnr;
            }
            printf("  per workgroup and launch: codeword loop %.0f .. %.0f s_memtime ticks, %.3f .. %.3f ms by s_memrealtime\n", mn, mx, mnr / 1e5, mxr / 1e5);
        }
        {
            std::vector<unsigned long long> wt(256 * 3);
            CK(hipMemcpyFromSymbol(wt.data(), HIP_SYMBOL(ldpc::g_wg_times), wt.size() * 8));
            unsigned long long e0 = ~0ull, e1 = 0, l0 = 0, x0 = ~0ull, x1 = 0;
            for (int b = 0; b < 256; ++b) {
                e0 = wt[b * 3] < e0 ? wt[b * 3] : e0; e1 = wt[b * 3] > e1 ? wt[b * 3] : e1; l0 = wt[b * 3 + 1] > l0 ? wt[b * 3 + 1] : l0;
                x0 = wt[b * 3 + 2] < x0 ? wt[b * 3 + 2] : x0; x1 = wt[b * 3 + 2] > x1 ? wt[b * 3 + 2] : x1;
            }
            printf("  last launch, s_memrealtime relative to the first workgroup's entry: last entry %.3f ms, last loop start %.3f ms, first exit %.3f ms, last exit %.3f ms\n",
                   (e1 - e0) / 1e5, (l0 - e0) / 1e5, (x0 - e0) / 1e5, (x1 - e0) / 1e5);
        }
        {   // per wave of the workgroup (wave w runs on SIMD w % 4; waves 4q .. 4q+3 own quarter q)
            double pw[16][4] = {};
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) for (int k = 0; k < 4; ++k) pw[w][k] += st[(b * 16 + w) * 6 + k];
            for (int w = 0; w < 16; ++w)
                printf("  wave %2d (SIMD %d): variable %4.0f | wait2 %4.0f | check %4.0f | wait1 %4.0f\n", w, w % 4, pw[w][0] / 256 / (iters_total / 256),
                       pw[w][1] / 256 / (iters_total / 256), pw[w][2] / 256 / (iters_total / 256), pw[w][3] / 256 / (iters_total / 256));
        }
        {
            std::vector<unsigned long long> ex(256 * 16 * 4);
            CK(hipMemcpyFromSymbol(ex.data(), HIP_SYMBOL(ldpc::g_stamp_ext), ex.size() * 8));
            for (int w = 0; w < 16; w += 4) {
                unsigned long long cmx = 0, cmn = ~0ull, wmx = 0, wmn = ~0ull;
                for (int b = 0; b < 256; ++b) {
                    const unsigned long long *x = &ex[(b * 16 + w) * 4];
                    cmx = x[0] > cmx ? x[0] : cmx; cmn = (x[1] && x[1] < cmn) ? x[1] : cmn; wmx = x[2] > wmx ? x[2] : wmx; wmn = (x[3] && x[3] < wmn) ? x[3] : wmn;
                }
                printf("  wave %2d over all workgroups and iterations: check phase %llu .. %llu ticks, wait behind it %llu .. %llu\n", w, cmn, cmx, wmn, wmx);
            }
        }
#if KPAIR
        {   // round 5: the fixed part of a codeword, per quarter of the workgroup's waves
            std::vector<unsigned long long> fx(256 * 16 * 8);
            CK(hipMemcpyFromSymbol(fx.data(), HIP_SYMBOL(ldpc::g_fixed), fx.size() * 8));
            static const char *name[7] = {"epilogue", "barrier behind it", "begin_codeword (zeroing, LLR wait, vote)", "barrier before the iterations",
                                          "pass 0: variable phase", "pass 0: barrier", "pass 0: check phase + ticket"};
            double all[8] = {};
            for (int q = 0; q < 4; ++q) {
                double sq[8] = {};
                for (int b = 0; b < 256; ++b) for (int w = 4 * q; w < 4 * q + 4; ++w) for (int k = 0; k < 8; ++k) sq[k] += fx[(b * 16 + w) * 8 + k];
                printf("  fixed part, quarter %d waves (ticks per codeword):", q);
                for (int k = 0; k < 7; ++k) { printf(" %5.0f", sq[k] / sq[7]); all[k] += sq[k]; }
                all[7] += sq[7];
                printf("\n");
            }
            double sum = 0;
            for (int k = 0; k < 7; ++k) { printf("  fixed part: %-42s %6.0f ticks per codeword\n", name[k], all[k] / all[7]); sum += all[k] / all[7]; }
            printf("  fixed part: total %.0f ticks per codeword (the stamped iterations above exclude pass 0)\n", sum);
        }
#endif
        for (int q = 0; q < 4; ++q)
            printf("  quarter %d waves: variable %.0f | wait2 %.0f | check %.0f | wait1 %.0f\n", q, perq[q][0] / 1024 / (iters_total / 256), perq[q][1] / 1024 / (iters_total / 256),
                   perq[q][2] / 1024 / (iters_total / 256), perq[q][3] / 1024 / (iters_total / 256));
    }
#endif
    if (claim) { uint32_t left = 1; CK(hipMemcpy(&left, claim, 4, hipMemcpyDeviceToHost)); if (left != 0) printf("QUEUE HEAD NOT RESET: %u\n", left); }
    printf("%scode %d T%zu ipt %d pf %d grid %u frames %zu: %.3f ms -> %.3f M cw/s | mean iters %.3f success %.5f | hash %016llx\n", KPAIR ? "PAIR " : "", code, sizeof(KT), KIPT, (int)KPF, groups, F, best,
           F / best / 1e3, si / F, sk / F, h);
    return 0;
}
