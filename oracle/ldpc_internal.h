/* ldpc_internal.h -- shared between the oracle's translation units. TEST INFRASTRUCTURE. */
#ifndef LDPC_INTERNAL_H
#define LDPC_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

struct edge_table {
    size_t n_edges;          /* E = paritycheck_sum (src/codes/mod.rs:85-86) */
    const uint16_t *check;   /* row index of edge e, in iter_paritychecks() order */
    const uint16_t *var;     /* column index of edge e */
};

/* NULL for an out-of-range code id. Thread-safe, built on first use. */
const struct edge_table *oracle_internal_edges(int code);

#endif
