// encoder.cpp -- host-side systematic encoder and edge-stream helpers.
//
// Behaviour provided: LDPCCode::encode / copy_encode (/root/reference/src/encoder.rs:293-315,
// C entries capi/src/lib.rs:25-46): data in the first k/8 bytes, parity written to the rest.
// The reference multiplies by stored generator circulants (src/codes/compact_generators.rs);
// this build stores none: the parity columns of H form an invertible square matrix for all
// nine codes, so the systematic generator is unique and is derived from H once per process.
// Only one row per row-of-circulants is solved for; the others are that row with every
// circulant_size-bit block rotated (the code is quasi-cyclic), which is what the reference's
// rotate-by-one loop at src/encoder.rs:72-80 exploits as well.
#include "host_codes.hpp"

#include <cstring>
#include <mutex>

namespace ldpc {

uint32_t edge_crc(int code)
{
    // CRC-32 (reflected 0xEDB88320, init all-ones, no final xor) fed check then var as 16-bit
    // steps: the checksum of test_iter_parity, /root/reference/src/codes/mod.rs:508-533.
    uint32_t crc = 0xFFFFFFFFu;
    auto feed = [&crc](uint32_t x) {
        crc ^= x;
        for (int i = 0; i < 16; ++i) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    };
    for_each_edge(code, [&](int chk, int var) { feed((uint32_t)chk); feed((uint32_t)var); });
    return crc;
}

namespace {

struct BitRows {
    size_t rows, words;
    std::vector<uint64_t> w;
    BitRows(size_t r, size_t bits) : rows(r), words((bits + 63) / 64), w(r * ((bits + 63) / 64), 0) {}
    uint64_t *row(size_t r) { return w.data() + r * words; }
    bool get(size_t r, size_t c) const { return (w[r * words + (c >> 6)] >> (c & 63)) & 1; }
    void flip(size_t r, size_t c) { w[r * words + (c >> 6)] ^= 1ull << (c & 63); }
};

Generator build_generator(int code)
{
    const CodeInfo &ci = CODES[code];
    const int k = ci.k, b = ci.circulant, C = ci.n_checks(), R = k / b, npar = ci.n - ci.k;

    // [ H_parity | H_data e_0 | H_data e_b | ... ]  -> reduce the left part to the identity
    BitRows aug(C, C + R);
    for_each_edge(code, [&](int chk, int var) {
        if (var >= k) aug.flip(chk, var - k);
        else if (var % b == 0) aug.flip(chk, C + var / b);
    });
    for (int col = 0; col < C; ++col) {
        int piv = col;
        while (piv < C && !aug.get(piv, col)) ++piv;
        if (piv == C) return Generator{};          // singular: no systematic form
        if (piv != col)
            for (size_t w = 0; w < aug.words; ++w) std::swap(aug.row(piv)[w], aug.row(col)[w]);
        const uint64_t *p = aug.row(col);
        for (int r = 0; r < C; ++r) {
            if (r == col || !aug.get(r, col)) continue;
            uint64_t *q = aug.row(r);
            for (size_t w = col >> 6; w < aug.words; ++w) q[w] ^= p[w];
        }
    }

    Generator g;
    g.k = k;
    g.parity_bytes = npar / 8;
    g.rows.assign((size_t)k * g.parity_bytes, 0);
    for (int crow = 0; crow < R; ++crow)
        for (int i = 0; i < npar; ++i) {
            if (!aug.get(i, C + crow)) continue;
            // parity bit i of data bit crow*b appears, for data bit crow*b + t, at the same
            // block but position (i + t) mod b
            const int blk0 = i - i % b, pos = i % b;
            for (int t = 0; t < b; ++t) {
                const int j = blk0 + ((pos + t) % b);
                g.rows[(size_t)(crow * b + t) * g.parity_bytes + j / 8] |= (uint8_t)(0x80u >> (j % 8));
            }
        }
    return g;
}

}  // namespace

const Generator *generator(int code)
{
    static Generator gens[NUM_CODES];
    static std::once_flag once[NUM_CODES];
    if (!valid_code(code)) return nullptr;
    std::call_once(once[code], [code] { gens[code] = build_generator(code); });
    return gens[code].k ? &gens[code] : nullptr;
}

void encode_parity(int code, const uint8_t *data, uint8_t *parity)
{
    const Generator *g = generator(code);
    if (!g) return;
    const int pb = g->parity_bytes;
    std::vector<uint8_t> acc(pb, 0);
    for (int d = 0; d < g->k; ++d) {
        if (!((data[d / 8] >> (7 - d % 8)) & 1)) continue;
        const uint8_t *row = g->rows.data() + (size_t)d * pb;
        for (int j = 0; j < pb; ++j) acc[j] ^= row[j];
    }
    std::memcpy(parity, acc.data(), pb);
}

}  // namespace ldpc
