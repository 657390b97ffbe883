/* ldpc_codes.c -- code definitions and the parity-check edge enumeration.
 *
 * TEST INFRASTRUCTURE (see ldpc_oracle.h).  Restates, in plain C:
 *   - `enum LDPCCode` / `CodeParams`          src/codes/mod.rs:37-66, :69-241
 *   - the prototype matrices, theta_k, phi_k  src/codes/compact_parity_checks.rs:17-268
 *     (CCSDS 231.1-O-1 for TC, CCSDS 131.0-B-2 for TM -- standard data)
 *   - the edge order of ParityIter::next      src/codes/mod.rs:275-362
 *   - iter_paritychecks_{tc,tm} selection     src/codes/mod.rs:435-494
 *
 * The reference stores each prototype as three byte-coded 4x11 layers that are
 * summed mod 2.  Here a prototype is written the way the CCSDS books print it:
 * one text cell per MxM sub-matrix, terms joined by '+':
 *     "-"   zero block                              (HZ, compact_parity_checks.rs:17)
 *     "Is"  identity circularly shifted right by s  (HI|s, :18; mod.rs:305-311)
 *     "Pk"  permutation pi_k, k = 1..26             (HP|(k-1), :110; mod.rs:312-322)
 * The order of terms inside a cell is the reference's layer order 0,1,2
 * (mod.rs:332-339), cells run left to right (mod.rs:343-350), rows top to
 * bottom (mod.rs:354-360) and the check index runs innermost (mod.rs:301-325):
 * that nesting IS the edge order which test_iter_parity's CRC pins.
 */
#include "ldpc_oracle.h"
#include "ldpc_internal.h"

#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ---- code parameters: src/codes/mod.rs:109-241 ------------------------------------- */
/* n, k, punctured bits, sub-matrix size M, circulant size, prototype id */
enum { PROTO_TC128, PROTO_TC256, PROTO_TC512, PROTO_TM_R12, PROTO_TM_R23, PROTO_TM_R45 };

static const struct { uint16_t n, k, p, m, circ; uint8_t proto; } PARAMS[ORACLE_NUM_CODES] = {
    /* TC128  */ {  128,   64,    0,   16,  16, PROTO_TC128  },
    /* TC256  */ {  256,  128,    0,   32,  32, PROTO_TC256  },
    /* TC512  */ {  512,  256,    0,   64,  64, PROTO_TC512  },
    /* TM1280 */ { 1280, 1024,  128,  128,  32, PROTO_TM_R45 },
    /* TM1536 */ { 1536, 1024,  256,  256,  64, PROTO_TM_R23 },
    /* TM2048 */ { 2048, 1024,  512,  512, 128, PROTO_TM_R12 },
    /* TM5120 */ { 5120, 4096,  512,  512, 128, PROTO_TM_R45 },
    /* TM6144 */ { 6144, 4096, 1024, 1024, 256, PROTO_TM_R23 },
    /* TM8192 */ { 8192, 4096, 2048, 2048, 512, PROTO_TM_R12 },
};

/* ---- prototype matrices: compact_parity_checks.rs:21-78 (TC), :113-170 (TM) -------- */
static const char *const PROTOS[6][4] = {
    [PROTO_TC128] = {
        "I0+I7 I2     I14    I6     -      I0     I13    I0",
        "I6    I0+I15 I0     I1     I0     -      I0     I7",
        "I4    I1     I0+I15 I14    I11    I0     -      I3",
        "I0    I1     I9     I0+I13 I14    I1     I0     -",
    },
    [PROTO_TC256] = {
        "I0+I31 I15    I25    I0     -      I20    I12    I0",
        "I28    I0+I30 I29    I24    I0     -      I1     I20",
        "I8     I0     I0+I28 I1     I29    I0     -      I21",
        "I18    I30    I0     I0+I30 I25    I26    I0     -",
    },
    [PROTO_TC512] = {
        "I0+I63 I30    I50    I25    -      I43    I62    I0",
        "I56    I0+I61 I50    I23    I0     -      I37    I26",
        "I16    I0     I0+I55 I27    I56    I0     -      I43",
        "I35    I56    I62    I0+I11 I58    I3     I0     -",
    },
    [PROTO_TM_R12] = {
        "-  -     I0 -     I0+P1",
        "I0 I0    -  I0    P2+P3+P4",
        "I0 P5+P6 -  P7+P8 I0",
        NULL,
    },
    [PROTO_TM_R23] = {
        "-          -           -  -     I0 -     I0+P1",
        "P9+P10+P11 I0          I0 I0    -  I0    P2+P3+P4",
        "I0         P12+P13+P14 I0 P5+P6 -  P7+P8 I0",
        NULL,
    },
    [PROTO_TM_R45] = {
        "-           -           -           -           -          -           -  -     I0 -     I0+P1",
        "P21+P22+P23 I0          P15+P16+P17 I0          P9+P10+P11 I0          I0 I0    -  I0    P2+P3+P4",
        "I0          P24+P25+P26 I0          P18+P19+P20 I0         P12+P13+P14 I0 P5+P6 -  P7+P8 I0",
        NULL,
    },
};

/* ---- theta_k and phi_k(j, M): compact_parity_checks.rs:174-244 ---------------------
 * Laid out as the CCSDS 131.0-B-2 tables print them: one row per k, then for each
 * j = 0..3 the values for M = 128, 256, 512, 1024, 2048.  (The M = 4096 / 8192
 * columns belong to the k=16384 codes, which the reference does not offer.) */
static const struct { uint8_t theta; uint16_t phi[4][5]; } PERM[26] = {
    /* k= 1 */ { 3, { {   1,  59,  16, 160, 108}, {   0,   0,   0,   0,   0}, {   0,   0,   0,   0,   0}, {   0,   0,   0,   0,   0} } },
    /* k= 2 */ { 0, { {  22,  18, 103, 241, 126}, {  27,  32,  53, 182, 375}, {  12,  46,   8,  35, 219}, {  13,  44,  35, 162, 312} } },
    /* k= 3 */ { 1, { {   0,  52, 105, 185, 238}, {  30,  21,  74, 249, 436}, {  30,  45, 119, 167,  16}, {  19,  51,  97,   7, 503} } },
    /* k= 4 */ { 2, { {  26,  23,   0, 251, 481}, {  28,  36,  45,  65, 350}, {  18,  27,  89, 214, 263}, {  14,  12, 112,  31, 388} } },
    /* k= 5 */ { 2, { {   0,  11,  50, 209,  96}, {   7,  30,  47,  70, 260}, {  10,  48,  31,  84, 415}, {  15,  15,  64, 164,  48} } },
    /* k= 6 */ { 3, { {  10,   7,  29, 103,  28}, {   1,  29,   0, 141,  84}, {  16,  37, 122, 206, 403}, {  20,  12,  93,  11,   7} } },
    /* k= 7 */ { 0, { {   5,  22, 115,  90,  59}, {   8,  44,  59, 237, 318}, {  13,  41,   1, 122, 184}, {  17,   4,  99, 237, 185} } },
    /* k= 8 */ { 1, { {  18,  25,  30, 184, 225}, {  20,  29, 102,  77, 382}, {   9,  13,  69,  67, 279}, {   4,   7,  94, 125, 328} } },
    /* k= 9 */ { 0, { {   3,  27,  92, 248, 323}, {  26,  39,  25,  55, 169}, {   7,   9,  92, 147, 198}, {   4,   2, 103, 133, 254} } },
    /* k=10 */ { 1, { {  22,  30,  78,  12,  28}, {  24,  14,   3,  12, 213}, {  15,  49,  47,  54, 307}, {  11,  30,  91,  99, 202} } },
    /* k=11 */ { 2, { {   3,  43,  70, 111, 386}, {   4,  22,  88, 227,  67}, {  16,  36,  11,  23, 432}, {  17,  53,   3, 105, 285} } },
    /* k=12 */ { 0, { {   8,  14,  66,  66, 305}, {  12,  15,  65,  42, 313}, {  18,  10,  31,  93, 240}, {  20,  23,   6,  17,  11} } },
    /* k=13 */ { 2, { {  25,  46,  39, 173,  34}, {  23,  48,  62,  52, 242}, {   4,  11,  19,  20, 454}, {   8,  29,  39,  97, 168} } },
    /* k=14 */ { 3, { {  25,  62,  84,  42, 510}, {  15,  55,  68, 243, 188}, {  23,  18,  66, 197, 294}, {  22,  37, 113,  91, 127} } },
    /* k=15 */ { 0, { {   2,  44,  79, 157, 147}, {  15,  39,  91, 179,   1}, {   5,  54,  49,  46, 479}, {  19,  42,  92, 211,   8} } },
    /* k=16 */ { 1, { {  27,  12,  70, 174, 199}, {  22,  11,  70, 250, 306}, {   3,  40,  81, 162, 289}, {  15,  48, 119, 128, 437} } },
    /* k=17 */ { 2, { {   7,  38,  29, 104, 347}, {  31,   1, 115, 247, 397}, {  29,  27,  96, 101, 373}, {   5,   4,  74,  82, 475} } },
    /* k=18 */ { 0, { {   7,  47,  32, 144, 391}, {   3,  50,  31, 164,  80}, {  11,  35,  38,  76, 104}, {  21,  10,  73, 115,  85} } },
    /* k=19 */ { 1, { {  15,   1,  45,  43, 165}, {  29,  40, 121,  17,  33}, {   4,  25,  83,  78, 141}, {  17,  18, 116, 248, 419} } },
    /* k=20 */ { 2, { {  10,  52, 113, 181, 414}, {  21,  62,  45,  31,   7}, {   8,  46,  42, 253, 270}, {   9,  56,  31,  62, 459} } },
    /* k=21 */ { 0, { {   4,  61,  86, 250,  97}, {   2,  27,  56, 149, 447}, {   2,  24,  58, 124, 439}, {  20,   9, 127,  26, 468} } },
    /* k=22 */ { 1, { {  19,  10,   1, 202, 158}, {   5,  38,  54, 105, 336}, {  11,  33,  24, 143, 333}, {  18,  11,  98, 140, 209} } },
    /* k=23 */ { 2, { {   7,  55,  42,  68,  86}, {  11,  40, 108, 183, 424}, {  11,  18,  25,  63, 399}, {  31,  23,  23, 121, 311} } },
    /* k=24 */ { 1, { {   9,   7, 118, 177, 168}, {  26,  15,  14, 153, 134}, {   3,  37,  92,  41,  14}, {  13,   8,  38,  12, 211} } },
    /* k=25 */ { 2, { {  26,  12,  33, 170, 506}, {   9,  11,  30, 177, 152}, {  15,  35,  38, 214, 277}, {   2,   7,  18,  41, 510} } },
    /* k=26 */ { 3, { {  17,   2, 126,  89, 489}, {  17,  18, 116,  19, 492}, {  13,  21, 120,  70, 412}, {  18,  24,  62, 249, 320} } },
};

static int phi_column(size_t m)   /* which M column of PERM[].phi: mod.rs:469-478 */
{
    switch (m) { case 128: return 0; case 256: return 1; case 512: return 2;
                 case 1024: return 3; case 2048: return 4; default: return -1; }
}

/* pi_k(i) = M/4 * ((theta_k + floor(4i/M)) mod 4) + (phi_k(floor(4i/M), M) + i) mod M/4
 * compact_parity_checks.rs:107-108, evaluated as in mod.rs:313-317. */
static size_t pi_k(int k, size_t i, size_t m)
{
    const size_t q = m / 4;
    const size_t j = i / q;
    const size_t phi = PERM[k - 1].phi[j][phi_column(m)];
    return q * ((PERM[k - 1].theta + j) % 4) + ((phi + i) % q);
}

/* ---- accessors: src/codes/mod.rs:367-409, src/decoder.rs:93-116 -------------------- */
#define BAD(code) ((code) < 0 || (code) >= ORACLE_NUM_CODES)

size_t oracle_code_n(int c)               { return BAD(c) ? 0 : PARAMS[c].n; }
size_t oracle_code_k(int c)               { return BAD(c) ? 0 : PARAMS[c].k; }
size_t oracle_code_punctured_bits(int c)  { return BAD(c) ? 0 : PARAMS[c].p; }
size_t oracle_code_submatrix_size(int c)  { return BAD(c) ? 0 : PARAMS[c].m; }
size_t oracle_code_circulant_size(int c)  { return BAD(c) ? 0 : PARAMS[c].circ; }

size_t oracle_bf_working_len(int c)    { return BAD(c) ? 0 : (size_t)PARAMS[c].n + PARAMS[c].p; }
size_t oracle_ms_working_u8_len(int c) { return BAD(c) ? 0 : ((size_t)PARAMS[c].n + PARAMS[c].p - PARAMS[c].k) / 8; }
size_t oracle_output_len(int c)        { return BAD(c) ? 0 : ((size_t)PARAMS[c].n + PARAMS[c].p) / 8; }
size_t oracle_ms_working_len(int c)
{
    if (BAD(c)) return 0;
    return 2 * oracle_code_paritycheck_sum(c) + 3 * (size_t)PARAMS[c].n + 3 * (size_t)PARAMS[c].p
           - 2 * (size_t)PARAMS[c].k;
}

/* ---- edge enumeration ----------------------------------------------------------------
 * Built once per code and cached; decode_ms walks it twice per iteration
 * (decoder.rs:388, :419).  The reference recomputes each pair on the fly from the
 * compact tables; tabulating them changes nothing observable. */
static struct edge_table TABLES[ORACLE_NUM_CODES];
static pthread_once_t TABLES_ONCE = PTHREAD_ONCE_INIT;

static void build_table(int code)
{
    const size_t m = PARAMS[code].m;
    const char *const *rows = PROTOS[PARAMS[code].proto];
    /* upper bound: 4 rows x 11 cols x 3 terms x M edges */
    size_t cap = 4 * 11 * 3 * m, e = 0;
    uint16_t *chk = malloc(cap * sizeof *chk), *var = malloc(cap * sizeof *var);
    if (!chk || !var) abort();

    for (size_t row = 0; row < 4 && rows[row]; row++) {         /* mod.rs:354-360 */
        const char *s = rows[row];
        size_t col = 0;
        while (*s) {                                            /* mod.rs:343-350 */
            while (*s == ' ') s++;
            if (!*s) break;
            /* one cell: terms joined by '+' -- the reference's layers 0,1,2 (mod.rs:332-339) */
            for (;;) {
                char kind = *s++;
                if (kind == '-') break;
                size_t val = strtoul(s, (char **)&s, 10);
                for (size_t i = 0; i < m; i++) {                /* mod.rs:301-325 */
                    size_t x = (kind == 'I') ? ((i + val) % m)  /* mod.rs:305-311 */
                                             : pi_k((int)val, i, m); /* mod.rs:312-322 */
                    chk[e] = (uint16_t)(row * m + i);
                    var[e] = (uint16_t)(col * m + x);
                    e++;
                }
                if (*s != '+') break;
                s++;
            }
            col++;
        }
    }
    TABLES[code].n_edges = e;
    TABLES[code].check = chk;
    TABLES[code].var = var;
}

static void build_all_tables(void)
{
    for (int c = 0; c < ORACLE_NUM_CODES; c++) build_table(c);
}

const struct edge_table *oracle_internal_edges(int code)
{
    if (BAD(code)) return NULL;
    pthread_once(&TABLES_ONCE, build_all_tables);
    return &TABLES[code];
}

size_t oracle_code_paritycheck_sum(int code)
{
    const struct edge_table *t = oracle_internal_edges(code);
    return t ? t->n_edges : 0;
}

size_t oracle_edges(int code, uint16_t *checks, uint16_t *vars, size_t cap)
{
    const struct edge_table *t = oracle_internal_edges(code);
    if (!t) return 0;
    size_t n = t->n_edges < cap ? t->n_edges : cap;
    if (checks) memcpy(checks, t->check, n * sizeof *checks);
    if (vars)   memcpy(vars,   t->var,   n * sizeof *vars);
    return t->n_edges;
}

/* CRC of test_iter_parity: src/codes/mod.rs:508-515 (update), :526-531 (driver). */
static uint32_t crc32_u16(uint32_t crc, uint32_t data)
{
    crc ^= data;
    for (int i = 0; i < 16; i++)
        crc = (crc >> 1) ^ ((crc & 1) ? 0xEDB88320u : 0u);
    return crc;
}

uint32_t oracle_edge_crc(int code)
{
    const struct edge_table *t = oracle_internal_edges(code);
    if (!t) return 0;
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t e = 0; e < t->n_edges; e++) {
        crc = crc32_u16(crc, t->check[e]);
        crc = crc32_u16(crc, t->var[e]);
    }
    return crc;
}
