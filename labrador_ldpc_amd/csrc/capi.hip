// capi.hip -- extern "C" entry points of liblabrador_ldpc_hip.so (include/labrador_ldpc_hip.h).
//
// Mirrors the reference's C API (/root/reference/capi/src/lib.rs:15-179) and adds the batched
// calls.  decode_ms always runs on the GPU; every failure is reported, never papered over
// with a CPU path.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>

#include "../../include/labrador_ldpc_hip.h"
#include "channel.hpp"
#include "encode.hpp"
#include "decode_bf.hpp"
#include "codes.hpp"
#include "host_codes.hpp"

namespace ldpc {
template <class T>
hipError_t launch_decode_ms(int code, int variant, const T *llrs, uint8_t *output, uint32_t *iters,
                            uint8_t *success, size_t batch, uint32_t maxiters, hipStream_t stream);
}

namespace {

thread_local std::string g_err;

int fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return status;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(LABRADOR_LDPC_HIP_ERUNTIME, "%s: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

bool device_is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

// Select the device the call should run on; returns a status and the previous device so the
// caller's context is left as found.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    int enter(const labrador_ldpc_hip_opts *opts)
    {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            (void)hipGetLastError();
            return fail(LABRADOR_LDPC_HIP_ENODEV, "no HIP device available (decode_ms has no CPU path)");
        }
        if (hipGetDevice(&prev) != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "hipGetDevice failed");
        int want = (opts && opts->device >= 0) ? opts->device : prev;
        if (want >= count) return fail(LABRADOR_LDPC_HIP_EINVAL, "device %d out of range (%d devices)", want, count);
        if (!device_is_gfx950(want))
            return fail(LABRADOR_LDPC_HIP_ENODEV, "device %d is not gfx950; this library carries gfx950 code only", want);
        if (want != prev) {
            if (hipSetDevice(want) != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "hipSetDevice(%d) failed", want);
            switched = true;
        }
        return LABRADOR_LDPC_HIP_OK;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

// Staging memory for the host-pointer entry points.  Each calling thread keeps a small set of
// grow-only device buffers per device, so the reference-shaped single-frame calls do not pay a
// hipMalloc/hipFree pair per frame (they are freed when the thread exits).
struct StagingPool {
    struct Slot { void *p = nullptr; size_t cap = 0; int dev = -1; };
    Slot slots[8];
    ~StagingPool() { for (auto &s : slots) if (s.p) (void)hipFree(s.p); }
    hipError_t get(int idx, size_t bytes, void **out)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        Slot &s = slots[idx];
        if (s.p && (s.dev != dev || s.cap < bytes)) {
            (void)hipFree(s.p);                       // implicit sync: no work of ours still uses it (calls are synchronous)
            s.p = nullptr; s.cap = 0;
        }
        if (!s.p) {
            const size_t cap = bytes < 4096 ? 4096 : bytes;
            e = hipMalloc(&s.p, cap);
            if (e != hipSuccess) { s.p = nullptr; return e; }
            s.cap = cap; s.dev = dev;
        }
        *out = s.p;
        return hipSuccess;
    }
};
thread_local StagingPool g_pool;

struct DeviceBuffer {                                 // a view into the calling thread's staging pool
    void *p = nullptr;
    int idx;
    explicit DeviceBuffer(int slot) : idx(slot) {}
    hipError_t alloc(size_t bytes) { return g_pool.get(idx, bytes, &p); }
};

template <class T>
int decode_batch(int code, const T *llrs, uint8_t *output, uint32_t *iters, uint8_t *success,
                 size_t batch, size_t max_iters, const labrador_ldpc_hip_opts *opts)
{
    g_err.clear();
    if (!ldpc::valid_code(code)) return fail(LABRADOR_LDPC_HIP_EINVAL, "code %d out of range", code);
    if (batch == 0) return LABRADOR_LDPC_HIP_OK;
    if (!llrs || !output || !iters || !success) return fail(LABRADOR_LDPC_HIP_EINVAL, "NULL buffer");
    const uint32_t maxit = max_iters > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)max_iters;
    const ldpc::CodeInfo &ci = ldpc::CODES[code];
    const size_t n = ci.n, out_len = ci.output_len();
    const int variant = opts ? opts->variant : 0;
    hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;

    DeviceScope scope;
    if (int s = scope.enter(opts)) return s;

    if (opts && opts->memory == LABRADOR_LDPC_HIP_MEM_DEVICE) {
        if ((uintptr_t)output % 8) return fail(LABRADOR_LDPC_HIP_EINVAL, "device output buffer must be 8-byte aligned");
        hipError_t e = ldpc::launch_decode_ms<T>(code, variant, llrs, output, iters, success, batch, maxit, stream);
        if (e == hipErrorInvalidConfiguration)
            return fail(LABRADOR_LDPC_HIP_EUNSUPPORTED, "kernel variant %d not built for code %d", variant, code);
        if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "kernel launch: %s", hipGetErrorString(e));
        return LABRADOR_LDPC_HIP_OK;
    }
    if (opts && opts->memory != LABRADOR_LDPC_HIP_MEM_HOST) return fail(LABRADOR_LDPC_HIP_EINVAL, "bad opts->memory");

    // host buffers: stage in chunks so that arbitrarily large batches fit
    const size_t chunk_max = (size_t)1 << 16;
    const size_t chunk = batch < chunk_max ? batch : chunk_max;
    DeviceBuffer d_llr(0), d_out(1), d_it(2), d_ok(3);
    HIP_TRY(d_llr.alloc(chunk * n * sizeof(T)));
    HIP_TRY(d_out.alloc(chunk * out_len));
    HIP_TRY(d_it.alloc(chunk * sizeof(uint32_t)));
    HIP_TRY(d_ok.alloc(chunk));
    for (size_t f0 = 0; f0 < batch; f0 += chunk) {
        const size_t nb = batch - f0 < chunk ? batch - f0 : chunk;
        HIP_TRY(hipMemcpyAsync(d_llr.p, llrs + f0 * n, nb * n * sizeof(T), hipMemcpyHostToDevice, stream));
        hipError_t e = ldpc::launch_decode_ms<T>(code, variant, (const T *)d_llr.p, (uint8_t *)d_out.p,
                                                 (uint32_t *)d_it.p, (uint8_t *)d_ok.p, nb, maxit, stream);
        if (e == hipErrorInvalidConfiguration)
            return fail(LABRADOR_LDPC_HIP_EUNSUPPORTED, "kernel variant %d not built for code %d", variant, code);
        if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "kernel launch: %s", hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(output + f0 * out_len, d_out.p, nb * out_len, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(iters + f0, d_it.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(success + f0, d_ok.p, nb, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return LABRADOR_LDPC_HIP_OK;
}

// capi/src/lib.rs:83-95: one frame, host pointers, optional iteration count
template <class T>
bool decode_one(int code, const T *llrs, uint8_t *output, size_t max_iters, size_t *iters_run)
{
    uint32_t it = 0;
    uint8_t ok = 0;
    const int s = decode_batch<T>(code, llrs, output, &it, &ok, 1, max_iters, nullptr);
    if (s != LABRADOR_LDPC_HIP_OK) {
        std::fprintf(stderr, "labrador_ldpc_hip: decode_ms failed: %s\n", g_err.c_str());
        return false;
    }
    // iters is clamped to 32 bits inside the kernel; report the caller's own bound on failure
    if (iters_run) *iters_run = ok ? (size_t)it : max_iters;
    return ok != 0;
}

// decoder.rs:484-493
template <class T>
void hard_to_llrs(int code, const uint8_t *input, T *llrs)
{
    if (!ldpc::valid_code(code)) return;
    const int n = ldpc::CODES[code].n;
    for (int i = 0; i < n; ++i) llrs[i] = ((input[i / 8] >> (7 - i % 8)) & 1) ? (T)-1 : (T)1;
}

// decoder.rs:498-509
template <class T>
void llrs_to_hard(int code, const T *llrs, uint8_t *output)
{
    if (!ldpc::valid_code(code)) return;
    const int n = ldpc::CODES[code].n;
    std::memset(output, 0, n / 8);
    for (int i = 0; i < n; ++i)
        if (llrs[i] < (T)0) output[i / 8] |= (uint8_t)(0x80u >> (i % 8));
}

template <class T>
int awgn(int code, const uint8_t *codewords, size_t pool, T *llrs, size_t batch, float sigma, float scale,
         int lim, uint64_t seed, const labrador_ldpc_hip_opts *opts)
{
    g_err.clear();
    if (!ldpc::valid_code(code)) return fail(LABRADOR_LDPC_HIP_EINVAL, "code %d out of range", code);
    if (batch == 0) return LABRADOR_LDPC_HIP_OK;
    if (!codewords || !llrs || pool == 0 || pool > 0xFFFFFFFFull) return fail(LABRADOR_LDPC_HIP_EINVAL, "bad codeword pool");
    if ((uintptr_t)llrs % 16) return fail(LABRADOR_LDPC_HIP_EINVAL, "llrs must be 16-byte aligned");
    DeviceScope scope;
    if (int s = scope.enter(opts)) return s;
    hipError_t e = ldpc::launch_awgn<T>(codewords, pool, llrs, ldpc::CODES[code].n, batch, sigma, scale, lim, seed,
                                        opts ? (hipStream_t)opts->stream : nullptr);
    if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "awgn launch: %s", hipGetErrorString(e));
    return LABRADOR_LDPC_HIP_OK;
}

}  // namespace

extern "C" {

// ---- sizes: capi/src/lib.rs:15-23, :48-66 ---------------------------------------------------
size_t labrador_ldpc_code_n(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].n : 0; }
size_t labrador_ldpc_code_k(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].k : 0; }
size_t labrador_ldpc_bf_working_len(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].bf_working_len() : 0; }
size_t labrador_ldpc_ms_working_len(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].ms_working_len() : 0; }
size_t labrador_ldpc_ms_working_u8_len(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].ms_working_u8_len() : 0; }
size_t labrador_ldpc_output_len(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::CODES[c].output_len() : 0; }

// ---- encoder: capi/src/lib.rs:25-46 ----------------------------------------------------------
void labrador_ldpc_encode(enum labrador_ldpc_code c, uint8_t *codeword)
{
    if (!ldpc::valid_code(c) || !codeword) return;
    ldpc::encode_parity(c, codeword, codeword + ldpc::CODES[c].k / 8);
}

void labrador_ldpc_copy_encode(enum labrador_ldpc_code c, const uint8_t *data, uint8_t *codeword)
{
    if (!ldpc::valid_code(c) || !data || !codeword) return;
    std::memmove(codeword, data, ldpc::CODES[c].k / 8);
    ldpc::encode_parity(c, codeword, codeword + ldpc::CODES[c].k / 8);
}

// ---- bit-flipping decoder: capi/src/lib.rs:68-81 -----------------------------------------------
int labrador_ldpc_decode_bf_batch(enum labrador_ldpc_code c, const uint8_t *input, uint8_t *output, uint32_t *iters,
                                  uint8_t *success, size_t batch, size_t max_iters,
                                  const struct labrador_ldpc_hip_opts *opts)
{
    g_err.clear();
    if (!ldpc::valid_code(c)) return fail(LABRADOR_LDPC_HIP_EINVAL, "code %d out of range", (int)c);
    if (batch == 0) return LABRADOR_LDPC_HIP_OK;
    if (!input || !output || !iters || !success) return fail(LABRADOR_LDPC_HIP_EINVAL, "NULL buffer");
    const uint32_t maxit = max_iters > 0x7FFFFFFEull ? 0x7FFFFFFEu : (uint32_t)max_iters;
    const size_t in_len = ldpc::CODES[c].n / 8, out_len = ldpc::CODES[c].output_len();
    hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;
    DeviceScope scope;
    if (int s = scope.enter(opts)) return s;
    if (opts && opts->memory == LABRADOR_LDPC_HIP_MEM_DEVICE) {
        hipError_t e = ldpc::launch_decode_bf(c, input, output, iters, success, batch, maxit, stream);
        if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "decode_bf launch: %s", hipGetErrorString(e));
        return LABRADOR_LDPC_HIP_OK;
    }
    if (opts && opts->memory != LABRADOR_LDPC_HIP_MEM_HOST) return fail(LABRADOR_LDPC_HIP_EINVAL, "bad opts->memory");
    DeviceBuffer d_in(0), d_out(1), d_it(2), d_ok(3);
    HIP_TRY(d_in.alloc(batch * in_len));
    HIP_TRY(d_out.alloc(batch * out_len));
    HIP_TRY(d_it.alloc(batch * sizeof(uint32_t)));
    HIP_TRY(d_ok.alloc(batch));
    HIP_TRY(hipMemcpyAsync(d_in.p, input, batch * in_len, hipMemcpyHostToDevice, stream));
    hipError_t e = ldpc::launch_decode_bf(c, (const uint8_t *)d_in.p, (uint8_t *)d_out.p, (uint32_t *)d_it.p,
                                          (uint8_t *)d_ok.p, batch, maxit, stream);
    if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "decode_bf launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(output, d_out.p, batch * out_len, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(iters, d_it.p, batch * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(success, d_ok.p, batch, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return LABRADOR_LDPC_HIP_OK;
}

bool labrador_ldpc_decode_bf(enum labrador_ldpc_code c, const uint8_t *input, uint8_t *output, uint8_t *,
                             size_t max_iters, size_t *iters_run)
{
    uint32_t it = 0;
    uint8_t ok = 0;
    const int s = labrador_ldpc_decode_bf_batch(c, input, output, &it, &ok, 1, max_iters, nullptr);
    if (s != LABRADOR_LDPC_HIP_OK) {
        std::fprintf(stderr, "labrador_ldpc_hip: decode_bf failed: %s\n", g_err.c_str());
        return false;
    }
    if (iters_run) *iters_run = ok ? (size_t)it : max_iters;
    return ok != 0;
}

// ---- min-sum decoder, one frame: capi/src/lib.rs:97-127 ---------------------------------------
bool labrador_ldpc_decode_ms_i8(enum labrador_ldpc_code c, const int8_t *llrs, uint8_t *output, int8_t *, uint8_t *,
                                size_t max_iters, size_t *iters_run)
{
    return decode_one<int8_t>(c, llrs, output, max_iters, iters_run);
}
bool labrador_ldpc_decode_ms_i16(enum labrador_ldpc_code c, const int16_t *llrs, uint8_t *output, int16_t *, uint8_t *,
                                 size_t max_iters, size_t *iters_run)
{
    return decode_one<int16_t>(c, llrs, output, max_iters, iters_run);
}
bool labrador_ldpc_decode_ms_f32(enum labrador_ldpc_code c, const float *llrs, uint8_t *output, float *, uint8_t *,
                                 size_t max_iters, size_t *iters_run)
{
    return decode_one<float>(c, llrs, output, max_iters, iters_run);
}
bool labrador_ldpc_decode_ms_f64(enum labrador_ldpc_code c, const double *llrs, uint8_t *output, double *, uint8_t *,
                                 size_t max_iters, size_t *iters_run)
{
    return decode_one<double>(c, llrs, output, max_iters, iters_run);
}

// ---- LLR helpers: capi/src/lib.rs:129-179 ------------------------------------------------------
void labrador_ldpc_hard_to_llrs_i8(enum labrador_ldpc_code c, const uint8_t *in, int8_t *llrs) { hard_to_llrs(c, in, llrs); }
void labrador_ldpc_hard_to_llrs_i16(enum labrador_ldpc_code c, const uint8_t *in, int16_t *llrs) { hard_to_llrs(c, in, llrs); }
void labrador_ldpc_hard_to_llrs_f32(enum labrador_ldpc_code c, const uint8_t *in, float *llrs) { hard_to_llrs(c, in, llrs); }
void labrador_ldpc_hard_to_llrs_f64(enum labrador_ldpc_code c, const uint8_t *in, double *llrs) { hard_to_llrs(c, in, llrs); }
void labrador_ldpc_llrs_to_hard_i8(enum labrador_ldpc_code c, const int8_t *llrs, uint8_t *out) { llrs_to_hard(c, llrs, out); }
void labrador_ldpc_llrs_to_hard_i16(enum labrador_ldpc_code c, const int16_t *llrs, uint8_t *out) { llrs_to_hard(c, llrs, out); }
void labrador_ldpc_llrs_to_hard_f32(enum labrador_ldpc_code c, const float *llrs, uint8_t *out) { llrs_to_hard(c, llrs, out); }
void labrador_ldpc_llrs_to_hard_f64(enum labrador_ldpc_code c, const double *llrs, uint8_t *out) { llrs_to_hard(c, llrs, out); }

// ---- batched GPU decoders ----------------------------------------------------------------------
int labrador_ldpc_decode_ms_batch_f32(enum labrador_ldpc_code c, const float *llrs, uint8_t *output, uint32_t *iters,
                                      uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts)
{
    return decode_batch<float>(c, llrs, output, iters, success, batch, max_iters, opts);
}
int labrador_ldpc_decode_ms_batch_i8(enum labrador_ldpc_code c, const int8_t *llrs, uint8_t *output, uint32_t *iters,
                                     uint8_t *success, size_t batch, size_t max_iters,
                                     const struct labrador_ldpc_hip_opts *opts)
{
    return decode_batch<int8_t>(c, llrs, output, iters, success, batch, max_iters, opts);
}
int labrador_ldpc_decode_ms_batch_i16(enum labrador_ldpc_code c, const int16_t *llrs, uint8_t *output, uint32_t *iters,
                                      uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts)
{
    return decode_batch<int16_t>(c, llrs, output, iters, success, batch, max_iters, opts);
}

// ---- batched encoder ------------------------------------------------------------------------------
int labrador_ldpc_encode_batch(enum labrador_ldpc_code c, const uint8_t *data, uint8_t *codewords, size_t batch,
                               const struct labrador_ldpc_hip_opts *opts)
{
    g_err.clear();
    if (!ldpc::valid_code(c)) return fail(LABRADOR_LDPC_HIP_EINVAL, "code %d out of range", (int)c);
    if (batch == 0) return LABRADOR_LDPC_HIP_OK;
    if (!data || !codewords) return fail(LABRADOR_LDPC_HIP_EINVAL, "NULL buffer");
    const size_t kb = ldpc::CODES[c].k / 8, nb = ldpc::CODES[c].n / 8;
    hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;
    DeviceScope scope;
    if (int s = scope.enter(opts)) return s;
    if (opts && opts->memory == LABRADOR_LDPC_HIP_MEM_DEVICE) {
        if ((uintptr_t)data % 4 || (uintptr_t)codewords % 4)
            return fail(LABRADOR_LDPC_HIP_EINVAL, "device buffers must be 4-byte aligned");
        hipError_t e = ldpc::launch_encode(c, data, codewords, batch, stream);
        if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "encode launch: %s", hipGetErrorString(e));
        return LABRADOR_LDPC_HIP_OK;
    }
    if (opts && opts->memory != LABRADOR_LDPC_HIP_MEM_HOST) return fail(LABRADOR_LDPC_HIP_EINVAL, "bad opts->memory");
    DeviceBuffer d_in(0), d_out(1);
    HIP_TRY(d_in.alloc(batch * kb));
    HIP_TRY(d_out.alloc(batch * nb));
    HIP_TRY(hipMemcpyAsync(d_in.p, data, batch * kb, hipMemcpyHostToDevice, stream));
    hipError_t e = ldpc::launch_encode(c, (const uint8_t *)d_in.p, (uint8_t *)d_out.p, batch, stream);
    if (e != hipSuccess) return fail(LABRADOR_LDPC_HIP_ERUNTIME, "encode launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(codewords, d_out.p, batch * nb, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return LABRADOR_LDPC_HIP_OK;
}

int labrador_ldpc_decode_ms_batch_f64(enum labrador_ldpc_code c, const double *llrs, uint8_t *output, uint32_t *iters,
                                      uint8_t *success, size_t batch, size_t max_iters,
                                      const struct labrador_ldpc_hip_opts *opts)
{
    return decode_batch<double>(c, llrs, output, iters, success, batch, max_iters, opts);
}

// ---- channel -------------------------------------------------------------------------------------
int labrador_ldpc_hip_awgn_f32(enum labrador_ldpc_code c, const uint8_t *codewords, size_t pool, float *llrs,
                               size_t batch, float sigma, uint64_t seed, const struct labrador_ldpc_hip_opts *opts)
{
    return awgn<float>(c, codewords, pool, llrs, batch, sigma, 1.0f, 0, seed, opts);
}
int labrador_ldpc_hip_awgn_i8(enum labrador_ldpc_code c, const uint8_t *codewords, size_t pool, int8_t *llrs,
                              size_t batch, float sigma, float scale, int lim, uint64_t seed,
                              const struct labrador_ldpc_hip_opts *opts)
{
    if (lim < 0 || lim > 127) return fail(LABRADOR_LDPC_HIP_EINVAL, "lim must be in 0..127");
    return awgn<int8_t>(c, codewords, pool, llrs, batch, sigma, scale, lim, seed, opts);
}

// ---- introspection ---------------------------------------------------------------------------------
uint32_t labrador_ldpc_hip_edge_crc(enum labrador_ldpc_code c) { return ldpc::valid_code(c) ? ldpc::edge_crc(c) : 0; }

int labrador_ldpc_hip_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int usable = 0;
    for (int d = 0; d < count; ++d) usable += device_is_gfx950(d) ? 1 : 0;
    return usable;
}

const char *labrador_ldpc_hip_last_error(void) { return g_err.c_str(); }
const char *labrador_ldpc_hip_version(void) { return "labrador_ldpc_hip 0.1.0 (gfx950)"; }

}  // extern "C"
