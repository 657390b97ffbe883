// encode.hip -- batched systematic encoder on the GPU.
//
// Batched form of LDPCCode::copy_encode (/root/reference/src/encoder.rs:293-315, the byte
// loop of :42-82): codeword = [ data | data * G_parity ] over GF(2), MSB-first bytes.
// The reference walks the generator row by row per set data bit.  Here a THREAD owns one
// parity column of the dense generator (k bits, held in VGPRs for the whole launch) and
// streams frames past it: a frame's data words are wave-uniform, so they arrive through the
// scalar cache, and one v_bitop3 per 32 data bits folds  acc ^= g & d.  The parity of the
// accumulator is the output bit; a wave ballot packs 64 of them into 8 output bytes.
// HBM traffic is the algorithmic minimum (k/8 bytes in, n/8 bytes out per frame); the
// generator (<= 2 MB) is read once per workgroup from L2.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "encode.hpp"
#include "host_codes.hpp"

namespace ldpc {

namespace {

// KW = k / 32 words per generator column
template <int KW>
__global__ void __launch_bounds__(256)
encode_kernel(const uint32_t *__restrict__ gt,     // [n-k][KW] generator columns, bit layout of a LE dword load
              const uint8_t *__restrict__ data,    // [batch][k/8]
              uint8_t *__restrict__ codewords,     // [batch][n/8]
              uint32_t batch, uint32_t n_bytes, uint32_t n_parity, uint32_t xcd_remap)
{
    constexpr int KB = KW * 4;                     // data bytes per frame
    // (column group cx, frame residue cy) of this workgroup: with xcd_remap the column groups of one frame run on one XCD (see
    // encode_kernel_k4096_2col below) -- one L2 fetches the frame's data and completes its parity lines
    uint32_t cx = blockIdx.x, cy = blockIdx.y;
    if (xcd_remap) {
        const uint32_t li = blockIdx.y * gridDim.x + blockIdx.x, slot = li >> 3;
        cx = slot % gridDim.x;
        cy = (slot / gridDim.x) * 8 + (li & 7);
    }
    const uint32_t p = cx * 256 + threadIdx.x;                  // parity column
    const bool active = p < n_parity;
    uint32_t g[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) g[w] = active ? gt[(size_t)p * KW + w] : 0u;

    for (uint32_t f = cy; f < batch; f += gridDim.y) {
        const uint32_t *d = reinterpret_cast<const uint32_t *>(data + (size_t)f * KB);   // wave-uniform
        // The data words are wave-uniform and arrive through the scalar cache in chunks of CH dwords; a chunk's
        // load is issued one chunk AHEAD of its use (two sets of SGPRs), otherwise every chunk costs the wave a
        // full scalar-load latency (~400 cycles against ~100 cycles of arithmetic per chunk).  Four independent
        // accumulators: one chain of KW dependent operations would leave the wave waiting on its own result.
        constexpr int CH = KW >= 32 ? 32 : KW, NCH = KW / CH, NA = KW >= 8 ? 4 : 1;
        uint32_t a4[NA] = {};
        uint32_t cur[CH], nxt[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[j] = d[j];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) {
#pragma unroll
                for (int j = 0; j < CH; ++j) nxt[j] = d[(c + 1) * CH + j];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j)
                a4[j % NA] = __builtin_amdgcn_bitop3_b32(a4[j % NA], g[c * CH + j], cur[j], 0x78);    // acc ^ (g & d)
            if (c + 1 < NCH) {
#pragma unroll
                for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
            }
        }
        uint32_t acc = a4[0];
#pragma unroll
        for (int i = 1; i < NA; ++i) acc ^= a4[i];
        const bool bit = __builtin_popcount(acc) & 1;
        const unsigned long long m = __ballot(bit);                                       // bit l = column p0 + l
        if ((threadIdx.x & 63) == 0 && active) {
            const unsigned lo = __builtin_bswap32(__builtin_bitreverse32((unsigned)m));
            const unsigned hi = __builtin_bswap32(__builtin_bitreverse32((unsigned)(m >> 32)));
            uint8_t *dst = codewords + (size_t)f * n_bytes + KB + p / 8;
            if (n_parity - p >= 64) {
                // 4-byte stores: n/8 and k/8 are multiples of 4 for every code, p/8 of 8
                reinterpret_cast<uint32_t *>(dst)[0] = lo;
                reinterpret_cast<uint32_t *>(dst)[1] = hi;
            } else {                               // TC128: 64 parity bits start mid-wave never happens; tail < 64 columns
                for (uint32_t b = 0; b < (n_parity - p + 7) / 8; ++b)
                    dst[b] = (uint8_t)((b < 4 ? lo >> (8 * b) : hi >> (8 * (b - 4))) & 0xFF);
            }
        }
        // systematic part: the first workgroup column copies the data bytes
        if (cx == 0)
            for (uint32_t b = threadIdx.x; b < (uint32_t)KW; b += 256)
                reinterpret_cast<uint32_t *>(codewords + (size_t)f * n_bytes)[b] = d[b];
    }
}

// k = 4096 (TM5120, TM6144, TM8192): the frame data reaches the arithmetic through LDS instead of the scalar
// cache.  A scalar load's latency (~400 cycles) cannot be hidden behind more than one chunk of arithmetic -- SMEM
// returns out of order, so a wave can only wait for ALL its outstanding loads -- which left the kernel above at a
// third of the VALU's rate for these codes.  Here the workgroup stages eight consecutive frames (4 KB, one 16-byte
// global load per thread, double-buffered, the next group's loads in flight during the arithmetic) and every wave
// reads a frame's words back as 32 broadcast ds_read_b128 (in-order, deeply pipelined by the hardware).
__global__ void __launch_bounds__(256, 2)
encode_kernel_k4096(const uint32_t *__restrict__ gt, const uint8_t *__restrict__ data, uint8_t *__restrict__ codewords,
                    uint32_t batch, uint32_t n_bytes, uint32_t n_parity, uint32_t frames_per_wg)
{
    constexpr int KW = 128, KB = 512, FB = 8;
    __shared__ uint4 stage[2][FB * KB / 16];
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x * 256 + tid;                  // parity column
    const bool active = p < n_parity;
    uint32_t g[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) g[w] = active ? gt[(size_t)p * KW + w] : 0u;

    const uint32_t f_begin = blockIdx.y * frames_per_wg;
    if (f_begin >= batch) return;                               // (uniform over the workgroup)
    const uint32_t f_end = f_begin + frames_per_wg < batch ? f_begin + frames_per_wg : batch;
    auto load_group = [&](uint32_t fg) -> uint4 {               // this thread's 16 bytes of the 8 frames from fg on
        if (fg + tid / 32 >= batch) return uint4{0u, 0u, 0u, 0u};
        return *reinterpret_cast<const uint4 *>(data + (size_t)fg * KB + tid * 16);
    };
    uint4 r = load_group(f_begin);
    int buf = 0;
    stage[0][tid] = r;
    __syncthreads();
    for (uint32_t fg = f_begin; fg < f_end; fg += FB) {
        const bool more = fg + FB < f_end;
        if (more) r = load_group(fg + FB);
        const uint32_t nf = f_end - fg < (uint32_t)FB ? f_end - fg : (uint32_t)FB;
        for (uint32_t j = 0; j < nf; ++j) {
            const uint4 *fr = &stage[buf][j * (KB / 16)];
            uint32_t a4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int w4 = 0; w4 < KW / 4; ++w4) {
                const uint4 dv = fr[w4];                                                  // same address in every lane
                a4[0] = __builtin_amdgcn_bitop3_b32(a4[0], g[4 * w4 + 0], dv.x, 0x78);  // acc ^ (g & d)
                a4[1] = __builtin_amdgcn_bitop3_b32(a4[1], g[4 * w4 + 1], dv.y, 0x78);
                a4[2] = __builtin_amdgcn_bitop3_b32(a4[2], g[4 * w4 + 2], dv.z, 0x78);
                a4[3] = __builtin_amdgcn_bitop3_b32(a4[3], g[4 * w4 + 3], dv.w, 0x78);
            }
            const uint32_t acc = a4[0] ^ a4[1] ^ a4[2] ^ a4[3];
            const unsigned long long m = __ballot(__builtin_popcount(acc) & 1);          // bit l = column p0 + l
            if ((tid & 63) == 0 && active) {                                              // n_parity is a multiple of 64 here
                const unsigned lo = __builtin_bswap32(__builtin_bitreverse32((unsigned)m));
                const unsigned hi = __builtin_bswap32(__builtin_bitreverse32((unsigned)(m >> 32)));
                uint32_t *dst = reinterpret_cast<uint32_t *>(codewords + (size_t)(fg + j) * n_bytes + KB + p / 8);
                dst[0] = lo;
                dst[1] = hi;
            }
        }
        // systematic part: the first column group copies the staged data bytes
        if (blockIdx.x == 0 && fg + tid / 32 < f_end)
            *reinterpret_cast<uint4 *>(codewords + (size_t)(fg + tid / 32) * n_bytes + (tid % 32) * 16) = stage[buf][tid];
        if (more) stage[buf ^ 1][tid] = r;
        __syncthreads();
        buf ^= 1;
    }
}

// k = 4096, round 4: TWO parity columns per thread over HALF of k.  The kernel above is bound by its LDS broadcasts, not by its
// arithmetic: a wave spends 32 ds_read_b128 (1.79 ns each on the CU's one LDS pipe: 57 ns) on the 128 v_bitop3 (141 ns on its SIMD) of a
// frame, and with eight waves per CU that is 458 ns of LDS against 282 ns of VALU per eight wave-frames (profiles/r04_final/frow_table.md:
// 49.8 M codewords/s = 71 % of that LDS bound).  A data word read once can feed two columns: threads 0-127 own columns (t, t + 128) of the
// workgroup's 256 over data words 0-63, threads 128-255 the same columns over words 64-127 -- the same 128 generator registers and the same
// 128 v_bitop3 per thread and frame, HALF the broadcasts.  The two halves' parities meet in LDS once per eight-frame group (the wave
// ballots, 1 KB, double-buffered behind the group loop's existing barrier) and 32 threads combine, bit-reverse and store them.
// Measured (profiles/r04_kbench/enc_variants.txt): 53.5 -> 62.4-64.5 M codewords/s TM8192, 190 -> 224-232 M TM5120.  Tried on top and
// dropped: 16- and 32-frame groups (-9..-12 %), four accumulators per column (+-1 %), issuing the broadcasts a 32-op chunk ahead
// (-5 %), the combine spread over the four waves (+-0.5 %).  What remains is the v_bitop3 issue itself: 128 per thread and frame.
__global__ void __launch_bounds__(256, 2)
encode_kernel_k4096_2col(const uint32_t *__restrict__ gt, const uint8_t *__restrict__ data, uint8_t *__restrict__ codewords,
                         uint32_t batch, uint32_t n_bytes, uint32_t n_parity, uint32_t frames_per_wg, uint32_t xcd_remap)
{
    constexpr int KW = 128, KB = 512, FB = 8, HW = KW / 2;
    // Which (column group cx, frame range cy) this workgroup takes.  Workgroups are dealt round-robin over the eight XCDs in launch
    // order (x fastest), so with the plain (blockIdx.x, blockIdx.y) assignment the gridDim.x column groups of ONE frame range sit on
    // different XCDs: each XCD's L2 fetches those frames' data for itself and writes its 32-byte pieces of their parity lines on its
    // own.  With the remap the workgroups that share an XCD (equal launch index mod 8) enumerate cx fastest, so one frame range's
    // column groups run side by side on one L2: the data is fetched once and the parity lines are completed there.  Placement is a
    // matter of memory traffic only (any bijection is correct, and the kernel's time does not depend on it: it is bound by its
    // v_bitop3 issue); it needs gridDim.y to be a multiple of 8.
    uint32_t cx = blockIdx.x, cy = blockIdx.y;
    if (xcd_remap) {
        const uint32_t li = blockIdx.y * gridDim.x + blockIdx.x, slot = li >> 3;
        cx = slot % gridDim.x;
        cy = (slot / gridDim.x) * 8 + (li & 7);
    }
    __shared__ uint4 stage[2][FB * KB / 16];
    __shared__ unsigned long long part[2][2][FB][2][2];         // [buffer][k half][frame][wave of the half][column set]
    const uint32_t tid = threadIdx.x;
    const uint32_t half = tid >> 7, t = tid & 127, wv = t >> 6;
    const uint32_t base = cx * 256;
    const uint32_t pa = base + t, pb = base + 128 + t;           // (n_parity is a multiple of 256 here)
    uint32_t ga[HW], gb[HW];
#pragma unroll
    for (int w = 0; w < HW; ++w) {
        ga[w] = gt[(size_t)pa * KW + half * HW + w];
        gb[w] = gt[(size_t)pb * KW + half * HW + w];
    }
    const uint32_t f_begin = cy * frames_per_wg;
    if (f_begin >= batch) return;                               // (uniform over the workgroup)
    const uint32_t f_end = f_begin + frames_per_wg < batch ? f_begin + frames_per_wg : batch;
    auto load_group = [&](uint32_t fg) -> uint4 {               // this thread's 16 bytes of the 8 frames from fg on
        if (fg + tid / 32 >= batch) return uint4{0u, 0u, 0u, 0u};
        return *reinterpret_cast<const uint4 *>(data + (size_t)fg * KB + tid * 16);
    };
    auto combine = [&](int b, uint32_t fg, uint32_t nf) {       // threads 0..31: one (frame, wave, column set) each
        if (tid < 32) {
            const uint32_t j = tid >> 2, w = (tid >> 1) & 1, sset = tid & 1;
            if (j < nf) {
                const unsigned long long m = part[b][0][j][w][sset] ^ part[b][1][j][w][sset];      // bit l = column base + 128 sset + 64 w + l
                const unsigned lo = __builtin_bswap32(__builtin_bitreverse32((unsigned)m));
                const unsigned hi = __builtin_bswap32(__builtin_bitreverse32((unsigned)(m >> 32)));
                uint32_t *dst = reinterpret_cast<uint32_t *>(codewords + (size_t)(fg + j) * n_bytes + KB + (base + 128 * sset + 64 * w) / 8);
                dst[0] = lo;
                dst[1] = hi;
            }
        }
    };
    uint4 r = load_group(f_begin);
    int buf = 0;
    stage[0][tid] = r;
    __syncthreads();
    uint32_t prev_fg = 0, prev_nf = 0;
    for (uint32_t fg = f_begin; fg < f_end; fg += FB) {
        const bool more = fg + FB < f_end;
        if (more) r = load_group(fg + FB);
        if (prev_nf) combine(buf ^ 1, prev_fg, prev_nf);       // the previous group's parities, complete since the barrier below
        const uint32_t nf = f_end - fg < (uint32_t)FB ? f_end - fg : (uint32_t)FB;
        for (uint32_t j = 0; j < nf; ++j) {
            const uint4 *fr = &stage[buf][j * (KB / 16) + half * (HW / 4)];
            uint32_t a2[2] = {0u, 0u}, b2[2] = {0u, 0u};
#pragma unroll
            for (int w4 = 0; w4 < HW / 4; ++w4) {
                const uint4 dv = fr[w4];                                                  // same address in every lane of a half
                a2[0] = __builtin_amdgcn_bitop3_b32(a2[0], ga[4 * w4 + 0], dv.x, 0x78);  // acc ^ (g & d)
                b2[0] = __builtin_amdgcn_bitop3_b32(b2[0], gb[4 * w4 + 0], dv.x, 0x78);
                a2[1] = __builtin_amdgcn_bitop3_b32(a2[1], ga[4 * w4 + 1], dv.y, 0x78);
                b2[1] = __builtin_amdgcn_bitop3_b32(b2[1], gb[4 * w4 + 1], dv.y, 0x78);
                a2[0] = __builtin_amdgcn_bitop3_b32(a2[0], ga[4 * w4 + 2], dv.z, 0x78);
                b2[0] = __builtin_amdgcn_bitop3_b32(b2[0], gb[4 * w4 + 2], dv.z, 0x78);
                a2[1] = __builtin_amdgcn_bitop3_b32(a2[1], ga[4 * w4 + 3], dv.w, 0x78);
                b2[1] = __builtin_amdgcn_bitop3_b32(b2[1], gb[4 * w4 + 3], dv.w, 0x78);
            }
            const unsigned long long ma = __ballot(__builtin_popcount(a2[0] ^ a2[1]) & 1);
            const unsigned long long mb = __ballot(__builtin_popcount(b2[0] ^ b2[1]) & 1);
            if ((tid & 63) == 0) { part[buf][half][j][wv][0] = ma; part[buf][half][j][wv][1] = mb; }
        }
        // systematic part: the first column group copies the staged data bytes
        if (cx == 0 && fg + tid / 32 < f_end)
            *reinterpret_cast<uint4 *>(codewords + (size_t)(fg + tid / 32) * n_bytes + (tid % 32) * 16) = stage[buf][tid];
        if (more) stage[buf ^ 1][tid] = r;
        __syncthreads();
        prev_fg = fg; prev_nf = nf;
        buf ^= 1;
    }
    if (prev_nf) combine(buf ^ 1, prev_fg, prev_nf);
}

struct DeviceGenerator {
    uint32_t *gt = nullptr;
};
DeviceGenerator g_tables[64][NUM_CODES];
std::mutex g_lock;

// column-major generator in the bit layout of little-endian dword loads of MSB-first bytes
hipError_t device_generator(int code, const uint32_t **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_lock);
    DeviceGenerator &t = g_tables[dev][code];
    if (!t.gt) {
        const Generator *gen = generator(code);
        if (!gen) return hipErrorInvalidValue;
        const CodeInfo &ci = CODES[code];
        const int k = ci.k, np = ci.n - ci.k, kw = k / 32;
        std::vector<uint32_t> host((size_t)np * kw, 0u);
        for (int d = 0; d < k; ++d) {
            const uint8_t *row = gen->rows.data() + (size_t)d * gen->parity_bytes;
            const int word = d / 32, bit = 8 * ((d / 8) % 4) + (7 - d % 8);
            for (int p = 0; p < np; ++p)
                if ((row[p / 8] >> (7 - p % 8)) & 1) host[(size_t)p * kw + word] |= 1u << bit;
        }
        e = hipMalloc(&t.gt, host.size() * sizeof(uint32_t));
        if (e != hipSuccess) return e;
        e = hipMemcpy(t.gt, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(t.gt); t.gt = nullptr; return e; }
    }
    *out = t.gt;
    return hipSuccess;
}

}  // namespace

hipError_t launch_encode(int code, const uint8_t *data, uint8_t *codewords, size_t batch, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const CodeInfo &ci = CODES[code];
    const uint32_t *gt = nullptr;
    hipError_t e = device_generator(code, &gt);
    if (e != hipSuccess) return e;
    const uint32_t np = ci.n - ci.k, nb = ci.n / 8;
    // grid.x = groups of 256 parity columns; grid.y = how many workgroups share the frames of one column
    // group.  Every workgroup first loads its 256 generator columns (k/8 bytes each: 128 KB for the k = 4096
    // codes) from L2, so grid.y is sized to the RESIDENT set -- a few times the CU count over grid.x -- rather
    // than to the batch: with one workgroup per handful of frames (round 1: min(batch, 4096)) the launch moved
    // 8.6 GB of generator for TM8192 whatever the batch, 3.3 ms of a 3.9 ms launch at 32 768 frames.
    const unsigned gx = (np + 255) / 256;
    // compute units of the CURRENT device (sharded workers and opts->device run on others than 0) and the kernels'
    // occupancy, queried once per device and kernel: a single-frame labrador_ldpc_encode-sized call should not pay two
    // uncached runtime queries per launch (the decode launchers cache the same quantity: resident_workgroups())
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static std::atomic<int> cached_cus[64] = {}, cached_occ[64][7] = {};         // [device][kernel]; concurrent fills store the same value
    int cus = cached_cus[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached_cus[dev].store(cus, std::memory_order_relaxed);
    }
    const void *kfn = nullptr;
    int kid = 0;
    switch (ci.k / 32) {
        case 2: kfn = (const void *)encode_kernel<2>; kid = 0; break;
        case 4: kfn = (const void *)encode_kernel<4>; kid = 1; break;
        case 8: kfn = (const void *)encode_kernel<8>; kid = 2; break;
        case 32: kfn = (const void *)encode_kernel<32>; kid = 3; break;
        case 128: kfn = (const void *)encode_kernel<128>; kid = 4; break;
        default: return hipErrorInvalidValue;
    }
    auto occupancy = [&](const void *fn, int slot) {
        int v = cached_occ[dev][slot].load(std::memory_order_relaxed);
        if (v == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, 256, 0) != hipSuccess || v < 1) v = 2;
            cached_occ[dev][slot].store(v, std::memory_order_relaxed);
        }
        return v;
    };
    const int per_cu = occupancy(kfn, kid);
    // exactly the resident set where a workgroup's generator load is heavy (k = 4096: 128 KB), several times it
    // for the small codes, whose workgroups are cheap to start and balance better when there are more of them
    const unsigned rounds = ci.k >= 4096 ? 1 : 4;
    unsigned gy = ((unsigned)(cus * per_cu) * rounds + gx - 1) / gx;
    if (ci.k >= 4096) gy = (unsigned)(cus * per_cu) / gx > 0 ? (unsigned)(cus * per_cu) / gx : 1;
    if (gy > batch) gy = (unsigned)batch;
    if (gy >= 8 && gx > 1) gy -= gy % 8;                         // the XCD-aware workgroup map needs a multiple of 8
    if (gy < 1) gy = 1;
    const dim3 grid(gx, gy);
    static const bool plain_map = [] { const char *e = std::getenv("LABRADOR_LDPC_HIP_ENC_PLAIN_MAP"); return e && *e && *e != '0'; }();
    const uint32_t remap_g = (!plain_map && gx > 1 && gy % 8 == 0) ? 1u : 0u;
    if (ci.k == 4096 && (uintptr_t)data % 16 == 0 && (uintptr_t)codewords % 16 == 0 && np % 256 == 0) {
        // LDS-staged kernels: contiguous runs of frames per workgroup, a multiple of their 8-frame stage.  Default: two columns per thread
        // over half of k (half the LDS broadcasts per v_bitop3); LABRADOR_LDPC_HIP_ENC_1COL=1 selects round 2's one column over all of k.
        static const bool one_col = [] { const char *e = std::getenv("LABRADOR_LDPC_HIP_ENC_1COL"); return e && *e && *e != '0'; }();
        if (!one_col) {
            const int pc = occupancy((const void *)encode_kernel_k4096_2col, 6);
            unsigned gy3 = (unsigned)(cus * pc) / gx > 0 ? (unsigned)(cus * pc) / gx : 1;
            size_t per_wg3 = (batch + gy3 - 1) / gy3;
            per_wg3 = (per_wg3 + 7) / 8 * 8;
            gy3 = (unsigned)((batch + per_wg3 - 1) / per_wg3);
            const uint32_t remap = (!plain_map && gy3 % 8 == 0) ? 1u : 0u;
            hipLaunchKernelGGL(encode_kernel_k4096_2col, dim3(gx, gy3), dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, (uint32_t)per_wg3, remap);
            return hipGetLastError();
        }
        const int per_cu2 = occupancy((const void *)encode_kernel_k4096, 5);
        unsigned gy2 = (unsigned)(cus * per_cu2) / gx > 0 ? (unsigned)(cus * per_cu2) / gx : 1;
        size_t per_wg = (batch + gy2 - 1) / gy2;
        per_wg = (per_wg + 7) / 8 * 8;
        gy2 = (unsigned)((batch + per_wg - 1) / per_wg);
        hipLaunchKernelGGL(encode_kernel_k4096, dim3(gx, gy2), dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, (uint32_t)per_wg);
        return hipGetLastError();
    }
    switch (ci.k / 32) {
        case 2:   hipLaunchKernelGGL((encode_kernel<2>),   grid, dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, remap_g); break;
        case 4:   hipLaunchKernelGGL((encode_kernel<4>),   grid, dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, remap_g); break;
        case 8:   hipLaunchKernelGGL((encode_kernel<8>),   grid, dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, remap_g); break;
        case 32:  hipLaunchKernelGGL((encode_kernel<32>),  grid, dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, remap_g); break;
        case 128: hipLaunchKernelGGL((encode_kernel<128>), grid, dim3(256), 0, stream, gt, data, codewords, (uint32_t)batch, nb, np, remap_g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ldpc
