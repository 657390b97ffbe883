/* Per-call latency of the reference-shaped single-frame entry from C (no binding in between): BASELINE config 1's shape -- TC128, one
 * frame, max_iters 50 -- and the same for TC512 / TM2048 / TM8192 f32 and i8, on frames that converge in a few iterations.
 *   cc -std=c11 -O2 -Iinclude tools/ubench/single_frame_latency.c -Llabrador_ldpc_amd -llabrador_ldpc_hip -L/opt/rocm/lib -lamdhip64 \
 *      -Wl,-rpath,$PWD/labrador_ldpc_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/sfl && /tmp/sfl */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "labrador_ldpc_hip.h"

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static int cmp(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

int main(void)
{
    const enum labrador_ldpc_code codes[] = {LABRADOR_LDPC_CODE_TC128, LABRADOR_LDPC_CODE_TC512, LABRADOR_LDPC_CODE_TM2048, LABRADOR_LDPC_CODE_TM8192};
    const char *names[] = {"TC128", "TC512", "TM2048", "TM8192"};
    static uint8_t data[512], cw[1024], out[1280];
    static float llr_f[8192];
    static int8_t llr_b[8192];
    printf("single-frame calls from C, us per call (median of 9 x 1000 calls), frames with ~3 %% of the bits flipped at low confidence\n");
    for (int c = 0; c < 4; ++c) {
        const size_t n = labrador_ldpc_code_n(codes[c]), k = labrador_ldpc_code_k(codes[c]);
        for (size_t i = 0; i < k / 8; ++i) data[i] = (uint8_t)(i * 37 + 11);
        labrador_ldpc_copy_encode(codes[c], data, cw);
        uint32_t s = 12345;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            float v = ((cw[i / 8] >> (7 - i % 8)) & 1) ? -4.0f : 4.0f;
            if ((s >> 8) % 33 == 0) v = -0.25f * v;
            llr_f[i] = v; llr_b[i] = (int8_t)(v * 8.0f);
        }
        double med[2];
        size_t iters = 0;
        for (int kind = 0; kind < 2; ++kind) {
            double t[9];
            for (int w = 0; w < 200; ++w) kind ? labrador_ldpc_decode_ms_i8(codes[c], llr_b, out, NULL, NULL, 50, &iters) : labrador_ldpc_decode_ms_f32(codes[c], llr_f, out, NULL, NULL, 50, &iters);
            for (int r = 0; r < 9; ++r) {
                const double a = now_us();
                for (int i = 0; i < 1000; ++i) {
                    const _Bool ok = kind ? labrador_ldpc_decode_ms_i8(codes[c], llr_b, out, NULL, NULL, 50, &iters)
                                          : labrador_ldpc_decode_ms_f32(codes[c], llr_f, out, NULL, NULL, 50, &iters);
                    if (!ok || memcmp(out, cw, n / 8) != 0) { printf("%s: wrong decode\n", names[c]); return 1; }
                }
                t[r] = (now_us() - a) / 1000;
            }
            qsort(t, 9, sizeof(double), cmp);
            med[kind] = t[4];
        }
        printf("%-7s f32 %6.2f   i8 %6.2f   (%zu iterations)\n", names[c], med[0], med[1], iters);
    }
    return 0;
}
