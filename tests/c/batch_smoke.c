/* C caller of the BATCHED part of the boundary (include/labrador_ldpc_hip.h, part 2), host buffers only --
 * the shape a non-Python host (the reference's Rust crate through extern "C", a C ground-station pipeline)
 * would use: encode a batch, corrupt it, decode it on one GPU, then again sharded over a device set, and
 * compare.  What perftest/src/main.rs:9-45 does per frame on CPU cores, as three library calls.
 *
 *   cc -Iinclude tests/c/batch_smoke.c -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 -o batch_smoke
 *   ./batch_smoke [frames]
 * Exit status 0 = all checks passed; 77 = no gfx950 device (nothing to run on: there is no CPU fallback).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "labrador_ldpc.h"

#ifndef CODE
#define CODE TM2048
#endif
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL: " __VA_ARGS__); printf(" [%s]\n", labrador_ldpc_hip_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const enum labrador_ldpc_code code = LABRADOR_LDPC_CODE(CODE);
    const size_t frames = argc > 1 ? (size_t)atol(argv[1]) : 3001;
    const size_t n = LABRADOR_LDPC_N(CODE), k = LABRADOR_LDPC_K(CODE), out_len = LABRADOR_LDPC_OUTPUT_LEN(CODE);
    if (labrador_ldpc_hip_device_count() == 0) { printf("no gfx950 device\n"); return 77; }

    uint8_t *data = malloc(frames * k / 8), *cws = malloc(frames * n / 8);
    int8_t *llrs = malloc(frames * n);
    uint8_t *out1 = malloc(frames * out_len), *out2 = malloc(frames * out_len), *ok1 = malloc(frames), *ok2 = malloc(frames);
    uint32_t *it1 = malloc(frames * 4), *it2 = malloc(frames * 4);
    CHECK(data && cws && llrs && out1 && out2 && ok1 && ok2 && it1 && it2, "malloc");
    unsigned s = 12345u;
    for (size_t i = 0; i < frames * k / 8; i++) { s = s * 1664525u + 1013904223u; data[i] = (uint8_t)(s >> 24); }

    /* batched encoder, default options (NULL = host memory, current device) */
    CHECK(labrador_ldpc_encode_batch(code, data, cws, frames, NULL) == LABRADOR_LDPC_HIP_OK, "encode_batch");
    uint8_t one[LABRADOR_LDPC_N(CODE) / 8];
    const size_t probe = frames > 7 ? 7 : 0;
    labrador_ldpc_copy_encode(code, data + probe * (k / 8), one);       /* the per-frame entry point agrees */
    CHECK(memcmp(one, cws + probe * (n / 8), n / 8) == 0, "encode_batch != copy_encode");

    /* +-8 LLRs with a burst of 6 weak wrong symbols per frame */
    for (size_t f = 0; f < frames; f++)
        for (size_t i = 0; i < n; i++) {
            const int bit = (cws[f * (n / 8) + i / 8] >> (7 - i % 8)) & 1;
            int v = bit ? -8 : 8;
            if (i >= 40 + f % 64 && i < 46 + f % 64) v = -v / 4;
            llrs[f * n + i] = (int8_t)v;
        }

    /* batched LLR helpers (host buffers): hard decisions of the LLRs differ from the codewords in exactly the burst */
    {
        int8_t *pm1 = malloc(frames * n);
        uint8_t *hard = malloc(frames * n / 8);
        CHECK(pm1 && hard, "malloc");
        CHECK(labrador_ldpc_hard_to_llrs_batch_i8(code, cws, pm1, frames, NULL) == LABRADOR_LDPC_HIP_OK, "hard_to_llrs_batch");
        CHECK(labrador_ldpc_llrs_to_hard_batch_i8(code, pm1, hard, frames, NULL) == LABRADOR_LDPC_HIP_OK, "llrs_to_hard_batch");
        CHECK(memcmp(hard, cws, frames * n / 8) == 0, "hard_to_llrs_batch / llrs_to_hard_batch do not round-trip");
        int8_t single_frame[LABRADOR_LDPC_N(CODE)];
        labrador_ldpc_hard_to_llrs_i8(code, cws + probe * (n / 8), single_frame);
        CHECK(memcmp(single_frame, pm1 + probe * n, n) == 0, "hard_to_llrs_batch != hard_to_llrs");
        CHECK(labrador_ldpc_llrs_to_hard_batch_i8(code, llrs, hard, frames, NULL) == LABRADOR_LDPC_HIP_OK, "llrs_to_hard_batch");
        size_t wrong = 0;
        for (size_t i = 0; i < frames * n / 8; i++) wrong += (size_t)__builtin_popcount(hard[i] ^ cws[i]);
        CHECK(wrong == 6 * frames, "%zu wrong hard decisions, expected %zu", wrong, 6 * frames);
        free(pm1); free(hard);
    }

    struct labrador_ldpc_hip_opts single = {0};                         /* device 0, host memory, default stream */
    CHECK(labrador_ldpc_decode_ms_batch_i8(code, llrs, out1, it1, ok1, frames, 25, &single) == 0, "decode on device 0");
    size_t good = 0;
    for (size_t f = 0; f < frames; f++) good += ok1[f] && memcmp(out1 + f * out_len, cws + f * (n / 8), n / 8) == 0;
    CHECK(good == frames, "only %zu of %zu frames decoded to the transmitted codeword", good, frames);

    /* the same batch sharded over every gfx950 device, then over an explicit list with a repeated ordinal */
    struct labrador_ldpc_hip_opts all = {0};
    all.device = LABRADOR_LDPC_HIP_DEVICE_ALL;
    CHECK(labrador_ldpc_decode_ms_batch_i8(code, llrs, out2, it2, ok2, frames, 25, &all) == 0, "decode on all devices");
    CHECK(!memcmp(out1, out2, frames * out_len) && !memcmp(it1, it2, frames * 4) && !memcmp(ok1, ok2, frames), "sharded != single");
    const int list[3] = {0, 0, 0};
    struct labrador_ldpc_hip_opts three = {0};
    three.n_devices = 3; three.devices = list;
    memset(out2, 0xEE, frames * out_len); memset(it2, 0xEE, frames * 4); memset(ok2, 0xEE, frames);
    CHECK(labrador_ldpc_decode_ms_batch_i8(code, llrs, out2, it2, ok2, frames, 25, &three) == 0, "decode on a device list");
    CHECK(!memcmp(out1, out2, frames * out_len) && !memcmp(it1, it2, frames * 4) && !memcmp(ok1, ok2, frames), "device list != single");
    size_t first, count;
    CHECK(labrador_ldpc_hip_shard_range(frames, 3, 2, &first, &count) == 0 && first + count == frames, "shard_range");

    /* errors come back as status codes with a text, never as a crash or a print */
    struct labrador_ldpc_hip_opts bad = {0};
    bad.device = 1000;
    CHECK(labrador_ldpc_decode_ms_batch_i8(code, llrs, out2, it2, ok2, 1, 25, &bad) == LABRADOR_LDPC_HIP_EINVAL, "bad device accepted");
    CHECK(strlen(labrador_ldpc_hip_last_error()) > 0, "no error text");

    printf("ok: %zu TM2048 frames encoded, decoded (i8, 25 iterations) on 1 device, on all %d, and on a 3-entry device list: identical\n",
           frames, labrador_ldpc_hip_device_count());
    return 0;
}
