/* ldpc_encode.c -- systematic encoder derived from the parity-check matrix.
 *
 * TEST INFRASTRUCTURE (see ldpc_oracle.h).  Behaviour restated: LDPCCode::encode /
 * copy_encode (src/encoder.rs:293-315): bytes [0, k/8) of the codeword are the data,
 * bytes [k/8, n/8) receive the parity, all bit-packed MSB first (src/encoder.rs:57).
 *
 * The reference multiplies by a table of generator circulants
 * (src/codes/compact_generators.rs, walked by src/encoder.rs:42-82).  That table is
 * not carried here.  H = [A | B] with B the (n+p-k) x (n+p-k) block over the parity
 * columns is full rank for all nine codes, so the systematic generator is unique and
 * is recovered by solving B x = A e for one unit data vector e per row of circulants
 * (k / circulant_size of them); all other rows follow from the quasi-cyclic structure:
 * shifting the data by t inside a circulant row shifts every circulant_size-bit block of
 * the parity by t (what src/encoder.rs:72-80 does with its rotate-by-one per offset).
 * Pinned by the 9 parity known-answers of src/encoder.rs:361-527.
 */
#include "ldpc_oracle.h"
#include "ldpc_internal.h"

#include <stdlib.h>
#include <string.h>
#include <pthread.h>

struct generator {
    int ready;            /* 0 = not built, 1 = ok, -1 = singular */
    size_t n_par;         /* C = n + p - k parity bits including punctured ones */
    size_t words;         /* u64 words per row */
    size_t rows;          /* k / circulant_size */
    uint64_t *x;          /* rows x words; bit i of a row = parity bit i for data bit row*b */
};

static struct generator GEN[ORACLE_NUM_CODES];
static pthread_mutex_t GEN_LOCK = PTHREAD_MUTEX_INITIALIZER;

static inline int  getbit(const uint64_t *w, size_t i) { return (int)((w[i >> 6] >> (i & 63)) & 1); }
static inline void flipbit(uint64_t *w, size_t i)      { w[i >> 6] ^= (uint64_t)1 << (i & 63); }

static void build_generator(int code)
{
    struct generator *g = &GEN[code];
    const struct edge_table *tab = oracle_internal_edges(code);
    const size_t n = oracle_code_n(code), k = oracle_code_k(code), p = oracle_code_punctured_bits(code);
    const size_t b = oracle_code_circulant_size(code);
    const size_t C = n + p - k, R = k / b;
    const size_t W = (C + R + 63) / 64;            /* augmented row: [B | A e_0 ... A e_{R-1}] */

    uint64_t *aug = calloc(C * W, sizeof *aug);
    if (!aug) abort();
    for (size_t e = 0; e < tab->n_edges; e++) {
        const size_t chk = tab->check[e], var = tab->var[e];
        if (var >= k)
            flipbit(aug + chk * W, var - k);                   /* B */
        else if (var % b == 0)
            flipbit(aug + chk * W, C + var / b);               /* A e_{var/b} */
    }

    /* Gauss-Jordan over GF(2) */
    int singular = 0;
    for (size_t col = 0; col < C && !singular; col++) {
        size_t piv = col;
        while (piv < C && !getbit(aug + piv * W, col)) piv++;
        if (piv == C) { singular = 1; break; }
        if (piv != col)
            for (size_t w = 0; w < W; w++) {
                uint64_t t = aug[piv * W + w]; aug[piv * W + w] = aug[col * W + w]; aug[col * W + w] = t;
            }
        const uint64_t *prow = aug + col * W;
        const size_t w0 = col >> 6;
        for (size_t r = 0; r < C; r++) {
            if (r == col || !getbit(aug + r * W, col)) continue;
            uint64_t *row = aug + r * W;
            for (size_t w = w0; w < W; w++) row[w] ^= prow[w];
        }
    }

    g->n_par = C;
    g->rows = R;
    g->words = (C + 63) / 64;
    if (singular) {
        g->ready = -1;
    } else {
        g->x = calloc(R * g->words, sizeof *g->x);
        if (!g->x) abort();
        for (size_t r = 0; r < R; r++)
            for (size_t i = 0; i < C; i++)
                if (getbit(aug + i * W, C + r))
                    flipbit(g->x + r * g->words, i);
        g->ready = 1;
    }
    free(aug);
}

static const struct generator *get_generator(int code)
{
    if (!oracle_internal_edges(code)) return NULL;
    pthread_mutex_lock(&GEN_LOCK);
    if (!GEN[code].ready) build_generator(code);
    pthread_mutex_unlock(&GEN_LOCK);
    return GEN[code].ready == 1 ? &GEN[code] : NULL;
}

/* rotate every b-bit block of `v` (C bits, bit i in word i/64 at position i%64) by one
 * position towards the higher index, wrapping inside the block */
static void rotate_blocks(uint64_t *v, size_t C, size_t b)
{
    if (b >= 64) {
        const size_t wpb = b / 64;
        for (size_t blk = 0; blk < C / b; blk++) {
            uint64_t *w = v + blk * wpb;
            uint64_t carry = w[wpb - 1] >> 63;
            for (size_t i = 0; i < wpb; i++) {
                uint64_t c = w[i] >> 63;
                w[i] = (w[i] << 1) | carry;
                carry = c;
            }
        }
    } else {
        const uint64_t field = ((uint64_t)1 << b) - 1;
        for (size_t blk = 0; blk < C / b; blk++) {
            const size_t pos = blk * b;
            uint64_t *w = v + (pos >> 6);
            const unsigned sh = (unsigned)(pos & 63);
            uint64_t f = (*w >> sh) & field;
            f = ((f << 1) | (f >> (b - 1))) & field;
            *w = (*w & ~(field << sh)) | (f << sh);
        }
    }
}

int oracle_encode(int code, uint8_t *codeword)
{
    const struct generator *g = get_generator(code);
    if (!g) return -1;
    const size_t n = oracle_code_n(code), k = oracle_code_k(code);
    const size_t b = oracle_code_circulant_size(code);

    uint64_t *par = calloc(g->words, sizeof *par);
    if (!par) abort();
    /* Horner over the offset inside a circulant: par = sum_t rot^t( sum_rows x[row] d[row*b+t] ) */
    for (size_t t = b; t-- > 0;) {
        rotate_blocks(par, g->n_par, b);
        for (size_t row = 0; row < g->rows; row++) {
            const size_t bit = row * b + t;
            if ((codeword[bit / 8] >> (7 - bit % 8)) & 1)
                for (size_t w = 0; w < g->words; w++) par[w] ^= g->x[row * g->words + w];
        }
    }
    /* transmitted parity = first n-k parity bits; the p punctured ones are dropped */
    memset(codeword + k / 8, 0, (n - k) / 8);
    for (size_t i = 0; i < n - k; i++)
        if (getbit(par, i))
            codeword[(k + i) / 8] |= (uint8_t)(1u << (7 - (k + i) % 8));
    free(par);
    return 0;
}

int oracle_copy_encode(int code, const uint8_t *data, uint8_t *codeword)
{
    const size_t k = oracle_code_k(code);
    if (!k) return -1;
    memcpy(codeword, data, k / 8);                 /* src/encoder.rs:100 */
    return oracle_encode(code, codeword);
}

size_t oracle_syndrome_weight(int code, const uint8_t *bits)
{
    const struct edge_table *tab = oracle_internal_edges(code);
    if (!tab) return (size_t)-1;
    const size_t C = oracle_code_n(code) + oracle_code_punctured_bits(code) - oracle_code_k(code);
    uint8_t *syn = calloc(C, 1);
    if (!syn) abort();
    for (size_t e = 0; e < tab->n_edges; e++) {
        const size_t var = tab->var[e];
        syn[tab->check[e]] ^= (bits[var / 8] >> (7 - var % 8)) & 1;
    }
    size_t wt = 0;
    for (size_t i = 0; i < C; i++) wt += syn[i];
    free(syn);
    return wt;
}
