/* Device-resident multi-GPU decode with ONE library call (round 4's review, item 7).
 *
 * tests/c/device_loop.c loops labrador_ldpc_decode_ms_batch_*(MEM_DEVICE) over the devices from the caller's own thread.  A host
 * that does not want to write that loop hands the library the per-device buffers instead:
 *
 *   labrador_ldpc_decode_ms_batch_{f32,i8}_multi(code, n_parts, devices[], llrs[], output[], iters[], success[], frames[], ...)
 *
 * -- part i is frames[i] frames resident on devices[i]; the library's persistent worker of each (device, occurrence) enqueues its
 * part on a stream of its own, and the call returns when every part is decoded (the reference's harness: one job, all workers,
 * perftest/src/main.rs:39-52).  This program checks that call against the one-call job on device 0:
 *
 *   job     = `frames` AWGN frames of CODE generated on the devices by GLOBAL frame index (shard [a, b) = bytes [a, b) of the
 *             one-device buffer), f32 and i8
 *   parts   = labrador_ldpc_hip_shard_range() slices over the device list: every gfx950 device, or argv[2] entries cycling over
 *             them ("3" with one GPU: three parts on device 0, three workers)
 *   check   = every part's outputs, iteration counts and flags equal its slice of the whole job, byte for byte; a part of zero
 *             frames and a bad device ordinal behave as the header says
 *
 *   cc -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/c/multi_call.c -Llabrador_ldpc_amd -llabrador_ldpc_hip \
 *      -L/opt/rocm/lib -lamdhip64 -o multi_call && ./multi_call [frames] [parts]
 * Exit status 0 = all checks passed; 77 = no gfx950 device.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "labrador_ldpc.h"

#ifndef CODE
#define CODE TM2048
#endif
#define MAX_PARTS 64
#define POOL 16
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL: " __VA_ARGS__); printf(" [%s]\n", labrador_ldpc_hip_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("FAIL: %s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static uint8_t pool_h[POOL * LABRADOR_LDPC_N(CODE) / 8];

/* the whole job on device 0 in one call; results to host */
static int whole_job(int i8, size_t frames, float sigma, uint64_t seed, uint8_t *w_out, uint32_t *w_it, uint8_t *w_ok)
{
    const enum labrador_ldpc_code code = LABRADOR_LDPC_CODE(CODE);
    const size_t n = LABRADOR_LDPC_N(CODE), out_len = LABRADOR_LDPC_OUTPUT_LEN(CODE), esz = i8 ? 1 : sizeof(float);
    HIP(hipSetDevice(0));
    uint8_t *d_pool, *d_out, *d_ok; void *d_llrs; uint32_t *d_it;
    HIP(hipMalloc((void **)&d_pool, sizeof pool_h));
    HIP(hipMalloc(&d_llrs, frames * n * esz));
    HIP(hipMalloc((void **)&d_out, frames * out_len));
    HIP(hipMalloc((void **)&d_it, frames * 4));
    HIP(hipMalloc((void **)&d_ok, frames));
    HIP(hipMemcpy(d_pool, pool_h, sizeof pool_h, hipMemcpyHostToDevice));
    struct labrador_ldpc_hip_opts o = LABRADOR_LDPC_HIP_OPTS_INIT;
    o.device = 0; o.memory = LABRADOR_LDPC_HIP_MEM_DEVICE;
    if (i8) {
        CHECK(labrador_ldpc_hip_awgn_i8(code, d_pool, POOL, (int8_t *)d_llrs, frames, sigma, 8.0f, 31, seed, &o) == 0, "awgn i8 (whole)");
        CHECK(labrador_ldpc_decode_ms_batch_i8(code, (const int8_t *)d_llrs, d_out, d_it, d_ok, frames, 25, &o) == 0, "decode i8 (whole)");
    } else {
        CHECK(labrador_ldpc_hip_awgn_f32(code, d_pool, POOL, (float *)d_llrs, frames, sigma, seed, &o) == 0, "awgn (whole)");
        CHECK(labrador_ldpc_decode_ms_batch_f32(code, (const float *)d_llrs, d_out, d_it, d_ok, frames, 25, &o) == 0, "decode (whole)");
    }
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(w_out, d_out, frames * out_len, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(w_it, d_it, frames * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(w_ok, d_ok, frames, hipMemcpyDeviceToHost));
    HIP(hipFree(d_pool)); HIP(hipFree(d_llrs)); HIP(hipFree(d_out)); HIP(hipFree(d_it)); HIP(hipFree(d_ok));
    return 0;
}

int main(int argc, char **argv)
{
    const enum labrador_ldpc_code code = LABRADOR_LDPC_CODE(CODE);
    const size_t frames = argc > 1 ? (size_t)atol(argv[1]) : 20011;
    const size_t n = LABRADOR_LDPC_N(CODE), k = LABRADOR_LDPC_K(CODE), out_len = LABRADOR_LDPC_OUTPUT_LEN(CODE);
    const int ndev = labrador_ldpc_hip_device_count();
    if (ndev == 0) { printf("no gfx950 device\n"); return 77; }
    int parts = argc > 2 ? atoi(argv[2]) : ndev;
    CHECK(parts >= 1 && parts < MAX_PARTS, "parts out of range");
    const float sigma = (k * 5 == n * 4) ? 0.5f : 0.75f;        /* Eb/N0 4 dB on the rate-4/5 codes, 2.5-3.7 dB on the others: most frames converge */
    const uint64_t seed = 0x1DBCull + (uint64_t)code;
    unsigned s = 2025u;
    for (int c = 0; c < POOL; c++) {
        uint8_t data[LABRADOR_LDPC_K(CODE) / 8];
        for (size_t i = 0; i < k / 8; i++) { s = s * 1664525u + 1013904223u; data[i] = (uint8_t)(s >> 24); }
        labrador_ldpc_copy_encode(code, data, pool_h + (size_t)c * (n / 8));
    }
    uint8_t *w_out = malloc(frames * out_len), *w_ok = malloc(frames);
    uint32_t *w_it = malloc(frames * 4);
    CHECK(w_out && w_ok && w_it, "malloc");

    for (int i8 = 0; i8 < 2; i8++) {
        const size_t esz = i8 ? 1 : sizeof(float);
        if (whole_job(i8, frames, sigma, seed, w_out, w_it, w_ok)) return 1;
        size_t converged = 0;
        for (size_t f = 0; f < frames; f++) converged += w_ok[f];
        CHECK(converged > frames / 2 && converged <= frames, "implausible job: %zu of %zu frames converged", converged, frames);

        /* the parts: [0, parts) = the shards of the job, + one EMPTY part at the end (frames = 0, NULL buffers: allowed) */
        int devices[MAX_PARTS];
        const void *llrs[MAX_PARTS];
        uint8_t *out[MAX_PARTS], *ok[MAX_PARTS], *pool[MAX_PARTS];
        uint32_t *it[MAX_PARTS];
        size_t first[MAX_PARTS], count[MAX_PARTS];
        for (int p = 0; p < parts; p++) {
            devices[p] = p % ndev;
            CHECK(labrador_ldpc_hip_shard_range(frames, (size_t)parts, (size_t)p, &first[p], &count[p]) == 0, "shard_range");
            HIP(hipSetDevice(devices[p]));
            void *l;
            HIP(hipMalloc((void **)&pool[p], sizeof pool_h));
            HIP(hipMalloc(&l, (count[p] ? count[p] : 1) * n * esz));
            HIP(hipMalloc((void **)&out[p], (count[p] ? count[p] : 1) * out_len));
            HIP(hipMalloc((void **)&it[p], (count[p] ? count[p] : 1) * 4));
            HIP(hipMalloc((void **)&ok[p], count[p] ? count[p] : 1));
            HIP(hipMemcpy(pool[p], pool_h, sizeof pool_h, hipMemcpyHostToDevice));
            struct labrador_ldpc_hip_opts o = LABRADOR_LDPC_HIP_OPTS_INIT;
            o.device = devices[p]; o.memory = LABRADOR_LDPC_HIP_MEM_DEVICE;
            if (i8) CHECK(labrador_ldpc_hip_awgn_i8_at(code, pool[p], POOL, (int8_t *)l, first[p], count[p], sigma, 8.0f, 31, seed, &o) == 0, "awgn_at i8 (part %d)", p);
            else CHECK(labrador_ldpc_hip_awgn_f32_at(code, pool[p], POOL, (float *)l, first[p], count[p], sigma, seed, &o) == 0, "awgn_at (part %d)", p);
            HIP(hipDeviceSynchronize());                 /* the inputs are complete before the multi call (its streams are the library's) */
            llrs[p] = l;
        }
        devices[parts] = 0; llrs[parts] = NULL; out[parts] = NULL; it[parts] = NULL; ok[parts] = NULL; count[parts] = 0;
        HIP(hipSetDevice(0));
        for (int round = 0; round < 2; round++) {        /* twice: the second call finds the workers and their streams warm */
            int st = i8 ? labrador_ldpc_decode_ms_batch_i8_multi(code, (size_t)parts + 1, devices, (const int8_t *const *)llrs, out, it, ok, count, 25, 0)
                        : labrador_ldpc_decode_ms_batch_f32_multi(code, (size_t)parts + 1, devices, (const float *const *)llrs, out, it, ok, count, 25, 0);
            CHECK(st == 0, "decode_ms_batch_%s_multi returned %d", i8 ? "i8" : "f32", st);
        }
        int cur = -1;
        HIP(hipGetDevice(&cur));
        CHECK(cur == 0, "the library left the calling thread on device %d", cur);
        size_t covered = 0;
        for (int p = 0; p < parts; p++) {
            HIP(hipSetDevice(devices[p]));
            CHECK(first[p] == covered, "parts are not contiguous");
            covered += count[p];
            uint8_t *o = malloc(count[p] * out_len + 1), *k1 = malloc(count[p] + 1);
            uint32_t *i1 = malloc(count[p] * 4 + 4);
            CHECK(o && k1 && i1, "malloc");
            /* no synchronisation here on purpose: the call returned, so the results must be in place */
            HIP(hipMemcpy(o, out[p], count[p] * out_len, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(i1, it[p], count[p] * 4, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(k1, ok[p], count[p], hipMemcpyDeviceToHost));
            CHECK(!memcmp(o, w_out + first[p] * out_len, count[p] * out_len), "%s part %d (device %d): outputs differ from the whole job's slice", i8 ? "i8" : "f32", p, devices[p]);
            CHECK(!memcmp(i1, w_it + first[p], count[p] * 4), "part %d: iteration counts differ", p);
            CHECK(!memcmp(k1, w_ok + first[p], count[p]), "part %d: success flags differ", p);
            free(o); free(k1); free(i1);
            HIP(hipFree(pool[p])); HIP(hipFree((void *)llrs[p])); HIP(hipFree(out[p])); HIP(hipFree(it[p])); HIP(hipFree(ok[p]));
        }
        CHECK(covered == frames, "parts cover %zu of %zu frames", covered, frames);
        printf("ok: %zu %s frames as %d device-resident parts (+ one empty) on %d device(s) through ONE call equal the one-call job (%zu converged)\n",
               frames, i8 ? "i8" : "f32", parts, ndev, converged);
    }
    /* argument errors are statuses, never crashes */
    {
        int bad_dev[1] = {ndev + 7};
        const float *l[1] = {(const float *)w_out};
        uint8_t *o[1] = {w_out}, *k1[1] = {w_ok};
        uint32_t *i1[1] = {w_it};
        size_t c[1] = {1};
        CHECK(labrador_ldpc_decode_ms_batch_f32_multi(code, 1, bad_dev, l, o, i1, k1, c, 25, 0) == LABRADOR_LDPC_HIP_EINVAL, "a device ordinal out of range must be EINVAL");
        CHECK(strstr(labrador_ldpc_hip_last_error(), "out of range") != NULL, "the error text names the problem");
        CHECK(labrador_ldpc_decode_ms_batch_f32_multi(code, 0, NULL, NULL, NULL, NULL, NULL, NULL, 25, 0) == 0, "zero parts is a no-op");
        CHECK(labrador_ldpc_decode_ms_batch_f32_multi(code, 1, NULL, l, o, i1, k1, c, 25, 0) == LABRADOR_LDPC_HIP_EINVAL, "NULL device list");
    }
    free(w_out); free(w_ok); free(w_it);
    return 0;
}
