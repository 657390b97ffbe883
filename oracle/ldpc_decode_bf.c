/* ldpc_decode_bf.c -- hard-decision decoders: erasure pre-pass and bit flipping.
 *
 * TEST INFRASTRUCTURE (see ldpc_oracle.h).  Restates, statement for statement:
 *   LDPCCode::decode_erasures   src/decoder.rs:144-223
 *   LDPCCode::decode_bf         src/decoder.rs:243-301
 * including their quirks: `bits_fixed` is incremented for every still-erased variable on every
 * iteration (src/decoder.rs:205-213), so the erasure pass returns (true, 0) after its first
 * iteration whenever maxiters >= 1; the parity bit of check i and the violation count of variable
 * i share byte i of the working area (src/decoder.rs:261-262).
 */
#include "ldpc_oracle.h"
#include "ldpc_internal.h"

#include <string.h>

/* src/decoder.rs:144-223.  Returns success; *iters_out receives the second tuple element. */
static int decode_erasures(int code, const struct edge_table *tab, uint8_t *codeword, uint8_t *working,
                           size_t maxiters, size_t *iters_out)
{
    const size_t n = oracle_code_n(code), p = oracle_code_punctured_bits(code);
    const size_t E = tab->n_edges;

    memset(working, 0x00, n);                                         /* :163 */
    memset(working + n, 0x10, p);                                     /* :164 */
    memset(codeword + n / 8, 0x00, p / 8);                            /* :167 */

    size_t bits_fixed = 0;                                            /* :170 */
    for (size_t iter = 0; iter < maxiters; iter++) {                  /* :172 */
        for (size_t i = 0; i < n + p; i++) working[i] = (uint8_t)((working[i] & 0x10) | 0x08);   /* :174 */

        for (size_t e = 0; e < E; e++) {                              /* :177-189 */
            const size_t check = tab->check[e], var = tab->var[e];
            if ((working[var] & 0x10) == 0x10) {
                switch (working[check] & 0x60) {
                    case 0x00: working[check] |= 0x20; break;
                    case 0x20: working[check] |= 0x40; break;
                    default: break;
                }
            } else if ((codeword[var / 8] >> (7 - (var % 8))) & 1) {
                working[check] ^= 0x80;
            }
        }
        for (size_t e = 0; e < E; e++) {                              /* :192-202 */
            const size_t check = tab->check[e], var = tab->var[e];
            if ((working[var] & 0x10) == 0x10 && (working[check] & 0x60) == 0x20) {
                if ((working[check] & 0x80) == 0x80) working[var] += 1;
                else working[var] -= 1;
            }
        }
        for (size_t var = 0; var < n + p; var++) {                    /* :205-213 */
            if ((working[var] & 0x10) == 0x10) {
                if ((working[var] & 0x0F) > 0x08) {
                    codeword[var / 8] |= (uint8_t)(1u << (7 - (var % 8)));
                    working[var] &= (uint8_t)~0x10;
                }
                bits_fixed += 1;
            }
        }
        if (bits_fixed == p) { *iters_out = iter; return 1; }         /* :215-218 */
    }
    *iters_out = maxiters;                                            /* :222 */
    return 0;
}

/* src/decoder.rs:243-301 */
int oracle_decode_bf(int code, const uint8_t *input, uint8_t *output, uint8_t *working,
                     size_t maxiters, size_t *iters_run)
{
    const struct edge_table *tab = oracle_internal_edges(code);
    if (!tab) return -1;
    const size_t n = oracle_code_n(code), p = oracle_code_punctured_bits(code);
    const size_t E = tab->n_edges;

    memcpy(output, input, n / 8);                                     /* :251 */

    size_t erasure_iters = 0;                                         /* :256-259 */
    if (p > 0) (void)decode_erasures(code, tab, output, working, maxiters, &erasure_iters);

    int success = 0;
    size_t iters = maxiters + erasure_iters;                          /* :300 */
    for (size_t iter = 0; iter < maxiters; iter++) {                  /* :264 */
        memset(working, 0, n + p);                                    /* :266 */
        for (size_t e = 0; e < E; e++) {                              /* :269-273 */
            const size_t var = tab->var[e];
            if ((output[var / 8] >> (7 - (var % 8))) & 1) working[tab->check[e]] ^= 0x80;
        }
        uint8_t max_violations = 0;                                   /* :276 */
        for (size_t e = 0; e < E; e++) {                              /* :277-286 */
            const size_t check = tab->check[e], var = tab->var[e];
            if ((working[check] & 0x80) == 0x80) {
                working[var] += 1;
                if ((working[var] & 0x7F) > max_violations) max_violations = working[var] & 0x7F;
            }
        }
        if (max_violations == 0) {                                    /* :288-289 */
            success = 1;
            iters = iter + erasure_iters;
            break;
        }
        for (size_t var = 0; var < n + p; var++)                      /* :292-296 */
            if ((working[var] & 0x7F) == max_violations)
                output[var / 8] ^= (uint8_t)(1u << (7 - (var % 8)));
    }
    if (iters_run) *iters_run = iters;
    return success;
}

/* exposed for tests: the erasure pre-pass alone, on a codeword buffer of output_len bytes */
int oracle_decode_erasures(int code, uint8_t *codeword, uint8_t *working, size_t maxiters, size_t *iters_run)
{
    const struct edge_table *tab = oracle_internal_edges(code);
    if (!tab) return -1;
    size_t it = 0;
    const int ok = decode_erasures(code, tab, codeword, working, maxiters, &it);
    if (iters_run) *iters_run = it;
    return ok;
}
