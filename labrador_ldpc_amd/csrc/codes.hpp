// codes.hpp -- compile-time description of the nine CCSDS codes for the HIP decoder.
//
// Replaces, for the product path, the reference's code registry and compact
// parity-check tables:
//   enum LDPCCode / CodeParams      /root/reference/src/codes/mod.rs:37-66, :69-241
//   prototype matrices, theta, phi  /root/reference/src/codes/compact_parity_checks.rs:17-268
//   ParityIter edge order           /root/reference/src/codes/mod.rs:275-362
//
// Everything here is constexpr so that the kernels see each sub-matrix ("block") as a
// set of literals: block row, block column, kind and shift are template-time constants,
// nothing is looked up at run time.  A prototype is written as the CCSDS books print it,
// rows separated by '/', one cell per MxM sub-matrix, terms of a cell joined by '+':
//     "-" zero, "Is" identity shifted right by s, "Pk" the permutation pi_k (k = 1..26).
// parse_prototype() flattens it into the block list in the reference's edge order:
// row-major over cells, terms of a cell in written order (the reference's three
// "layers"), and -- inside a block -- check index ascending.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define LDPC_HD __host__ __device__
#else
#define LDPC_HD
#endif

namespace ldpc {

enum : int { TC128 = 0, TC256, TC512, TM1280, TM1536, TM2048, TM5120, TM6144, TM8192, NUM_CODES };

enum BlockKind : uint8_t { BLK_I = 0, BLK_P = 1 };

struct Block {
    uint8_t row;    // block row    (check index = row * M + i)
    uint8_t col;    // block column (variable index = col * M + f(i))
    uint8_t kind;   // BLK_I: f(i) = (i + val) mod M ; BLK_P: f(i) = pi_val(i)
    uint8_t val;    // shift s, or k of pi_k (1-based as in CCSDS 131.0-B-2)
};

constexpr int MAX_BLOCKS = 40;

struct Prototype {
    Block blk[MAX_BLOCKS] = {};
    int n_blocks = 0;
    int n_rows = 0;     // block rows
    int n_cols = 0;     // block columns, punctured one included
};

constexpr Prototype parse_prototype(const char *s)
{
    Prototype p{};
    int row = 0, col = 0;
    while (*s) {
        if (*s == ' ') { ++s; continue; }
        if (*s == '/') { ++row; col = 0; ++s; continue; }
        // one cell
        for (;;) {
            const char kind = *s++;
            if (kind == '-') break;
            int val = 0;
            while (*s >= '0' && *s <= '9') val = val * 10 + (*s++ - '0');
            p.blk[p.n_blocks++] = Block{(uint8_t)row, (uint8_t)col,
                                        (uint8_t)(kind == 'P' ? BLK_P : BLK_I), (uint8_t)val};
            if (*s != '+') break;
            ++s;
        }
        ++col;
        if (col > p.n_cols) p.n_cols = col;
        p.n_rows = row + 1;
    }
    return p;
}

// CCSDS 231.1-O-1 (TC) -- compact_parity_checks.rs:21-78
inline constexpr Prototype PROTO_TC128 = parse_prototype(
    "I0+I7 I2     I14    I6     -   I0  I13 I0 /"
    "I6    I0+I15 I0     I1     I0  -   I0  I7 /"
    "I4    I1     I0+I15 I14    I11 I0  -   I3 /"
    "I0    I1     I9     I0+I13 I14 I1  I0  -");
inline constexpr Prototype PROTO_TC256 = parse_prototype(
    "I0+I31 I15    I25    I0     -   I20 I12 I0  /"
    "I28    I0+I30 I29    I24    I0  -   I1  I20 /"
    "I8     I0     I0+I28 I1     I29 I0  -   I21 /"
    "I18    I30    I0     I0+I30 I25 I26 I0  -");
inline constexpr Prototype PROTO_TC512 = parse_prototype(
    "I0+I63 I30    I50    I25    -   I43 I62 I0  /"
    "I56    I0+I61 I50    I23    I0  -   I37 I26 /"
    "I16    I0     I0+I55 I27    I56 I0  -   I43 /"
    "I35    I56    I62    I0+I11 I58 I3  I0  -");
// CCSDS 131.0-B-2 (TM), rates 1/2, 2/3, 4/5 -- compact_parity_checks.rs:113-170
inline constexpr Prototype PROTO_TM_R12 = parse_prototype(
    "-  -     I0 -     I0+P1    /"
    "I0 I0    -  I0    P2+P3+P4 /"
    "I0 P5+P6 -  P7+P8 I0");
inline constexpr Prototype PROTO_TM_R23 = parse_prototype(
    "-          -           -  -     I0 -     I0+P1    /"
    "P9+P10+P11 I0          I0 I0    -  I0    P2+P3+P4 /"
    "I0         P12+P13+P14 I0 P5+P6 -  P7+P8 I0");
inline constexpr Prototype PROTO_TM_R45 = parse_prototype(
    "-           -           -           -           -          -           -  -     I0 -     I0+P1    /"
    "P21+P22+P23 I0          P15+P16+P17 I0          P9+P10+P11 I0          I0 I0    -  I0    P2+P3+P4 /"
    "I0          P24+P25+P26 I0          P18+P19+P20 I0         P12+P13+P14 I0 P5+P6 -  P7+P8 I0");

// theta_k and phi_k(j, M) of CCSDS 131.0-B-2 (compact_parity_checks.rs:174-244), one row per
// k = 1..26; phi[j][c] with c = log2(M) - 7 for M = 128, 256, 512, 1024, 2048.
struct PermRow { uint8_t theta; uint16_t phi[4][5]; };
inline constexpr PermRow PERM[26] = {
    { 3, { {   1,  59,  16, 160, 108}, {   0,   0,   0,   0,   0}, {   0,   0,   0,   0,   0}, {   0,   0,   0,   0,   0} } },
    { 0, { {  22,  18, 103, 241, 126}, {  27,  32,  53, 182, 375}, {  12,  46,   8,  35, 219}, {  13,  44,  35, 162, 312} } },
    { 1, { {   0,  52, 105, 185, 238}, {  30,  21,  74, 249, 436}, {  30,  45, 119, 167,  16}, {  19,  51,  97,   7, 503} } },
    { 2, { {  26,  23,   0, 251, 481}, {  28,  36,  45,  65, 350}, {  18,  27,  89, 214, 263}, {  14,  12, 112,  31, 388} } },
    { 2, { {   0,  11,  50, 209,  96}, {   7,  30,  47,  70, 260}, {  10,  48,  31,  84, 415}, {  15,  15,  64, 164,  48} } },
    { 3, { {  10,   7,  29, 103,  28}, {   1,  29,   0, 141,  84}, {  16,  37, 122, 206, 403}, {  20,  12,  93,  11,   7} } },
    { 0, { {   5,  22, 115,  90,  59}, {   8,  44,  59, 237, 318}, {  13,  41,   1, 122, 184}, {  17,   4,  99, 237, 185} } },
    { 1, { {  18,  25,  30, 184, 225}, {  20,  29, 102,  77, 382}, {   9,  13,  69,  67, 279}, {   4,   7,  94, 125, 328} } },
    { 0, { {   3,  27,  92, 248, 323}, {  26,  39,  25,  55, 169}, {   7,   9,  92, 147, 198}, {   4,   2, 103, 133, 254} } },
    { 1, { {  22,  30,  78,  12,  28}, {  24,  14,   3,  12, 213}, {  15,  49,  47,  54, 307}, {  11,  30,  91,  99, 202} } },
    { 2, { {   3,  43,  70, 111, 386}, {   4,  22,  88, 227,  67}, {  16,  36,  11,  23, 432}, {  17,  53,   3, 105, 285} } },
    { 0, { {   8,  14,  66,  66, 305}, {  12,  15,  65,  42, 313}, {  18,  10,  31,  93, 240}, {  20,  23,   6,  17,  11} } },
    { 2, { {  25,  46,  39, 173,  34}, {  23,  48,  62,  52, 242}, {   4,  11,  19,  20, 454}, {   8,  29,  39,  97, 168} } },
    { 3, { {  25,  62,  84,  42, 510}, {  15,  55,  68, 243, 188}, {  23,  18,  66, 197, 294}, {  22,  37, 113,  91, 127} } },
    { 0, { {   2,  44,  79, 157, 147}, {  15,  39,  91, 179,   1}, {   5,  54,  49,  46, 479}, {  19,  42,  92, 211,   8} } },
    { 1, { {  27,  12,  70, 174, 199}, {  22,  11,  70, 250, 306}, {   3,  40,  81, 162, 289}, {  15,  48, 119, 128, 437} } },
    { 2, { {   7,  38,  29, 104, 347}, {  31,   1, 115, 247, 397}, {  29,  27,  96, 101, 373}, {   5,   4,  74,  82, 475} } },
    { 0, { {   7,  47,  32, 144, 391}, {   3,  50,  31, 164,  80}, {  11,  35,  38,  76, 104}, {  21,  10,  73, 115,  85} } },
    { 1, { {  15,   1,  45,  43, 165}, {  29,  40, 121,  17,  33}, {   4,  25,  83,  78, 141}, {  17,  18, 116, 248, 419} } },
    { 2, { {  10,  52, 113, 181, 414}, {  21,  62,  45,  31,   7}, {   8,  46,  42, 253, 270}, {   9,  56,  31,  62, 459} } },
    { 0, { {   4,  61,  86, 250,  97}, {   2,  27,  56, 149, 447}, {   2,  24,  58, 124, 439}, {  20,   9, 127,  26, 468} } },
    { 1, { {  19,  10,   1, 202, 158}, {   5,  38,  54, 105, 336}, {  11,  33,  24, 143, 333}, {  18,  11,  98, 140, 209} } },
    { 2, { {   7,  55,  42,  68,  86}, {  11,  40, 108, 183, 424}, {  11,  18,  25,  63, 399}, {  31,  23,  23, 121, 311} } },
    { 1, { {   9,   7, 118, 177, 168}, {  26,  15,  14, 153, 134}, {   3,  37,  92,  41,  14}, {  13,   8,  38,  12, 211} } },
    { 2, { {  26,  12,  33, 170, 506}, {   9,  11,  30, 177, 152}, {  15,  35,  38, 214, 277}, {   2,   7,  18,  41, 510} } },
    { 3, { {  17,   2, 126,  89, 489}, {  17,  18, 116,  19, 492}, {  13,  21, 120,  70, 412}, {  18,  24,  62, 249, 320} } },
};

constexpr int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// pi_k(i) = M/4 * ((theta_k + floor(4i/M)) mod 4) + (phi_k(floor(4i/M), M) + i) mod M/4
// (compact_parity_checks.rs:107-108; evaluated by the reference at mod.rs:313-317)
LDPC_HD constexpr int theta_of(int k) { return PERM[k - 1].theta; }
LDPC_HD constexpr int phi_of(int k, int j, int M) { return PERM[k - 1].phi[j][ilog2(M) - 7]; }
LDPC_HD constexpr int pi_k(int k, int i, int M)
{
    const int q = M / 4, j = i / q;
    return q * ((theta_of(k) + j) & 3) + ((phi_of(k, j, M) + i) & (q - 1));
}

// Variable index inside its block column that check `i` of block `b` is connected to.
LDPC_HD constexpr int block_map(const Block &b, int i, int M)
{
    return b.kind == BLK_I ? ((i + b.val) & (M - 1)) : pi_k(b.val, i, M);
}

// Per-code constants (mod.rs:109-241).  M = sub-matrix size; the block columns beyond
// n / M are the punctured ones.
struct CodeInfo {
    int n, k, p, m, circulant;
    const Prototype *proto;
    constexpr int n_checks() const { return n + p - k; }
    constexpr int n_edges() const { return proto->n_blocks * m; }           // paritycheck_sum
    constexpr int output_len() const { return (n + p) / 8; }                // decoder.rs:114-116
    constexpr int ms_working_len() const { return 2 * n_edges() + 3 * n + 3 * p - 2 * k; } // :100-102
    constexpr int ms_working_u8_len() const { return (n + p - k) / 8; }     // :107-109
    constexpr int bf_working_len() const { return n + p; }                  // :93-95
};

inline constexpr CodeInfo CODES[NUM_CODES] = {
    {  128,   64,    0,   16,  16, &PROTO_TC128  },
    {  256,  128,    0,   32,  32, &PROTO_TC256  },
    {  512,  256,    0,   64,  64, &PROTO_TC512  },
    { 1280, 1024,  128,  128,  32, &PROTO_TM_R45 },
    { 1536, 1024,  256,  256,  64, &PROTO_TM_R23 },
    { 2048, 1024,  512,  512, 128, &PROTO_TM_R12 },
    { 5120, 4096,  512,  512, 128, &PROTO_TM_R45 },
    { 6144, 4096, 1024, 1024, 256, &PROTO_TM_R23 },
    { 8192, 4096, 2048, 2048, 512, &PROTO_TM_R12 },
};

constexpr bool valid_code(int code) { return code >= 0 && code < NUM_CODES; }

}  // namespace ldpc
