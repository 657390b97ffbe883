/* C caller of the boundary, in the shape of the reference's only C client
 * (capi/examples/example.c:23-95): every buffer statically sized by the header's macros, then
 * copy_encode -> corrupt -> hard_to_llrs_f32 -> decode_ms_f32 (200 iterations max) -> compare.
 * Compiled per code with -DCODE=<name> by tests/test_c_boundary.py; the decode step needs a
 * gfx950 device (there is no CPU fallback), everything before it runs anywhere.
 *
 *   cc -Iinclude -DCODE=TM2048 tests/c/example_smoke.c -Llabrador_ldpc_amd -llabrador_ldpc_hip \
 *      -L/opt/rocm/lib -lamdhip64 -o example_smoke
 *
 * Exit status: 0 = all checks passed, 1 = a check failed, 77 = host-side checks passed but the
 * decode was skipped because no device is present (argument "--host-only").
 */
#include <stdio.h>
#include <string.h>
#include "labrador_ldpc.h"

#ifndef CODE
#define CODE TC128
#endif

static uint8_t data_bytes[LABRADOR_LDPC_K(CODE) / 8];
static uint8_t sent[LABRADOR_LDPC_N(CODE) / 8];
static uint8_t received[LABRADOR_LDPC_N(CODE) / 8];
static float   soft[LABRADOR_LDPC_N(CODE)];
static float   ms_work[LABRADOR_LDPC_MS_WORKING_LEN(CODE)];
static uint8_t ms_work_u8[LABRADOR_LDPC_MS_WORKING_U8_LEN(CODE)];
static uint8_t bf_work[LABRADOR_LDPC_BF_WORKING_LEN(CODE)];
static uint8_t decoded[LABRADOR_LDPC_OUTPUT_LEN(CODE)];

_Static_assert(LABRADOR_LDPC_OUTPUT_LEN(CODE) * 8 == LABRADOR_LDPC_BF_WORKING_LEN(CODE),
               "output_len is (n+p)/8, bf_working_len is n+p");

#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL: " __VA_ARGS__); printf("\n"); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int host_only = argc > 1 && strcmp(argv[1], "--host-only") == 0;
    const enum labrador_ldpc_code code = LABRADOR_LDPC_CODE(CODE);
    const size_t n = labrador_ldpc_code_n(code), k = labrador_ldpc_code_k(code);

    /* the static sizes are the run-time sizes */
    CHECK(n == LABRADOR_LDPC_N(CODE), "n: %zu vs macro %d", n, LABRADOR_LDPC_N(CODE));
    CHECK(k == LABRADOR_LDPC_K(CODE), "k: %zu vs macro %d", k, LABRADOR_LDPC_K(CODE));
    CHECK(labrador_ldpc_bf_working_len(code) == sizeof bf_work, "bf_working_len");
    CHECK(labrador_ldpc_ms_working_len(code) == sizeof ms_work / sizeof ms_work[0], "ms_working_len");
    CHECK(labrador_ldpc_ms_working_u8_len(code) == sizeof ms_work_u8, "ms_working_u8_len");
    CHECK(labrador_ldpc_output_len(code) == sizeof decoded, "output_len");

    for (size_t i = 0; i < k / 8; i++) data_bytes[i] = (uint8_t)(37u * i + 11u);
    labrador_ldpc_copy_encode(code, data_bytes, sent);
    CHECK(memcmp(sent, data_bytes, k / 8) == 0, "the encoder is systematic");

    /* in-place encode of the same data must give the same codeword */
    memcpy(received, data_bytes, k / 8);
    labrador_ldpc_encode(code, received);
    CHECK(memcmp(received, sent, n / 8) == 0, "encode and copy_encode disagree");

    /* corrupt: wipe the last data byte, as the reference example does (example.c:62) */
    received[k / 8 - 1] ^= 0xFF;
    labrador_ldpc_hard_to_llrs_f32(code, received, soft);
    for (size_t i = 0; i < n; i++) {
        const int bit = (received[i / 8] >> (7 - i % 8)) & 1;
        CHECK(soft[i] == (bit ? -1.0f : 1.0f), "hard_to_llrs_f32 at %zu", i);
    }
    uint8_t back[sizeof received];
    labrador_ldpc_llrs_to_hard_f32(code, soft, back);
    CHECK(memcmp(back, received, sizeof back) == 0, "llrs_to_hard_f32 does not invert hard_to_llrs_f32");

    if (host_only) { printf("host checks ok (%d: n=%zu k=%zu), decode skipped\n", (int)code, n, k); return 77; }

    size_t iters = (size_t)-1;
    const bool ok = labrador_ldpc_decode_ms_f32(code, soft, decoded, ms_work, ms_work_u8, 200, &iters);
    CHECK(ok, "decode_ms_f32 failed: %s", labrador_ldpc_hip_last_error());
    CHECK(iters < 200, "iterations %zu", iters);
    CHECK(memcmp(decoded, sent, n / 8) == 0, "decode_ms_f32 did not restore the codeword");

    /* the hard-decision decoder on the same word, single flipped bit */
    memcpy(received, sent, n / 8);
    received[0] ^= 0x10;
    size_t bf_iters = 0;
    const bool bf_ok = labrador_ldpc_decode_bf(code, received, decoded, bf_work, 50, &bf_iters);
    CHECK(bf_ok, "decode_bf failed: %s", labrador_ldpc_hip_last_error());
    CHECK(memcmp(decoded, sent, n / 8) == 0, "decode_bf did not restore the codeword");

    printf("ok: code %d n=%zu k=%zu, min-sum converged at iteration %zu, bit-flipping after %zu\n",
           (int)code, n, k, iters, bf_iters);
    return 0;
}
