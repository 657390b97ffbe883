// host_codes.hpp -- host-side helpers derived from the constexpr code tables: edge stream,
// systematic generator, LLR helpers.  Product code (not the oracle): it shares nothing
// with oracle/ except the published CCSDS constants.
#pragma once

#include <cstdint>
#include <cstddef>
#include <vector>

#include "codes.hpp"

namespace ldpc {

// Call f(check, var) for every edge in the reference's order (mod.rs:275-362): blocks in list
// order, check index ascending inside a block.
template <class F>
inline void for_each_edge(int code, F &&f)
{
    const CodeInfo &ci = CODES[code];
    for (int b = 0; b < ci.proto->n_blocks; ++b) {
        const Block &blk = ci.proto->blk[b];
        for (int i = 0; i < ci.m; ++i)
            f(blk.row * ci.m + i, blk.col * ci.m + block_map(blk, i, ci.m));
    }
}

uint32_t edge_crc(int code);

// Dense systematic generator, parity part only: row d (data bit d) holds the n-k transmitted
// parity bits, packed MSB-first in bytes -- the layout `codeword[k/8..]` uses.  Built on first
// use from H (see encoder.cpp); returns nullptr if H's parity part is singular.
struct Generator {
    int k = 0, parity_bytes = 0;
    std::vector<uint8_t> rows;     // k * parity_bytes
};
const Generator *generator(int code);

void encode_parity(int code, const uint8_t *data, uint8_t *parity);

}  // namespace ldpc
