"""ms_numpy.py -- a SECOND, independently structured CPU restatement of the reference's min-sum decoder.

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/: only tests/ and the golden-vector generator
may import it; nothing in the product does).

Purpose: the reference's own tests assert no iteration counts and no non-converged outputs
(src/decoder.rs:671-699 checks success + codeword only), so those are pinned by restatement fidelity.
To keep a transcription slip in the C oracle (oracle/ldpc_decode_tmpl.h, an edge-by-edge restatement
in the reference's own loop order) from going unnoticed, this file states the same algorithm
(src/decoder.rs:347-475, scalar semantics :22-86) in a different shape and from the reference text,
not from the C oracle:

  * whole-array operations over [frames, edges] instead of a loop over edges;
  * check-to-variable messages formed for all edges at once from the previous pass's per-check
    minima / sign products (:387-405);
  * marginals accumulated by "occurrence rank": rank r holds, for every variable, its r-th edge in
    edge order, so adding rank 0, 1, 2, ... reproduces the reference's sequential, non-associative
    accumulation order per variable (:408) without visiting edges one by one;
  * the strict-`<` two-minimum update (:430-434) replaced by its closed form: min1 / min2 are the
    smallest and second-smallest magnitude of the check's messages *as a multiset*, each capped at
    T::maxval() (they start there, :414-415, and only strictly smaller values replace them -- so an
    infinite magnitude never does), obtained by sorting a padded [frames, checks, max degree] array;
  * sign products and parities as XOR reductions over the same padded layout (:438-447).

The edge list (check, var) in iterator order is an input (tests pass the CRC-pinned list of
oracle.edges(); the CRCs are the reference's own, src/codes/mod.rs:517-535).
"""
from __future__ import annotations

import numpy as np

_INT_MAX = {np.dtype(np.int8): 127, np.dtype(np.int16): 32767, np.dtype(np.int32): 2147483647}


class Structure:
    """Index tables derived once per code from the ordered edge list."""

    def __init__(self, checks: np.ndarray, variables: np.ndarray, n_vars: int):
        checks = np.asarray(checks, dtype=np.int64)
        variables = np.asarray(variables, dtype=np.int64)
        self.E = len(checks)
        self.chk, self.var = checks, variables
        self.n_vars = n_vars                      # n + p
        self.n_checks = int(checks.max()) + 1
        # occurrence rank of every edge among the edges of its variable, in edge order
        seen = np.zeros(n_vars, dtype=np.int64)
        rank = np.empty(self.E, dtype=np.int64)
        for e in range(self.E):
            rank[e] = seen[variables[e]]
            seen[variables[e]] += 1
        self.rank_groups = [np.nonzero(rank == r)[0] for r in range(int(rank.max()) + 1)]
        # padded per-check edge table: pad entries point at a dummy edge slot E
        deg = np.bincount(checks, minlength=self.n_checks)
        self.maxdeg = int(deg.max())
        self.by_check = np.full((self.n_checks, self.maxdeg), self.E, dtype=np.int64)
        fill = np.zeros(self.n_checks, dtype=np.int64)
        for e in range(self.E):
            c = checks[e]
            self.by_check[c, fill[c]] = e
            fill[c] += 1


def decode_ms(st: Structure, llrs: np.ndarray, n: int, maxiters: int):
    """llrs [frames, n] of dtype i8/i16/i32/f32/f64 -> (output [frames, (n+p)/8] u8, iters [frames] u32,
    success [frames] u8), each frame exactly as LDPCCode::decode_ms would return it."""
    llrs = np.ascontiguousarray(llrs)
    dt = llrs.dtype
    F = llrs.shape[0]
    is_float = dt.kind == "f"
    if is_float:
        wt = dt                                   # arithmetic in the type itself: IEEE add/sub, nothing else
        maxval = np.finfo(dt).max
        pad = np.array(np.inf, dtype=dt)
        add = lambda a, b: a + b                  # noqa: E731   (:74-75)
        sub = lambda a, b: a - b                  # noqa: E731
        mag = np.abs                              # clears the sign bit (:73)
    else:
        wt = np.dtype(np.int64)                   # integers: exact in 64 bits, saturated back to the type
        hi, lo = _INT_MAX[dt], -_INT_MAX[dt] - 1
        maxval = hi
        pad = np.array(1 << 40, dtype=wt)
        add = lambda a, b: np.clip(a + b, lo, hi)  # noqa: E731   saturating_add (:47)
        sub = lambda a, b: np.clip(a - b, lo, hi)  # noqa: E731   saturating_sub (:48)
        mag = lambda a: np.minimum(np.abs(a), hi)  # noqa: E731   saturating_abs: |MIN| -> MAX (:46)
    E, C, V = st.E, st.n_checks, st.n_vars
    zero = np.zeros((), dtype=wt)

    out_len = V // 8
    output = np.zeros((F, out_len), dtype=np.uint8)
    iters = np.full(F, maxiters, dtype=np.uint32)
    success = np.zeros(F, dtype=np.uint8)

    # the reference zeroes the whole working area (:374): u, v, va, both minima; and the sign bitmap (:366)
    v = np.zeros((F, E), dtype=wt)
    min1 = np.zeros((F, C), dtype=wt)
    min2 = np.zeros((F, C), dtype=wt)
    sgn = np.zeros((F, C), dtype=bool)
    va = np.zeros((F, V), dtype=wt)
    live = np.arange(F)                           # frames still iterating (row i of the state = frame live[i])
    L = llrs.astype(wt)

    def hard_pack(vals):
        bits = (vals < zero).astype(np.uint8)     # hard_bit: strictly negative (:49, :76); -0.0 is not
        return np.packbits(bits, axis=1)          # MSB first (:459)

    for it in range(maxiters):
        if len(live) == 0:
            break
        # ---- pass 1 (:387-411): u for every edge from last pass's minima / signs and last v
        m1e, m2e = min1[:, st.chk], min2[:, st.chk]
        u = np.where(mag(v) == m1e, m2e, m1e)
        u = np.where(sgn[:, st.chk], -u, u)
        u = np.where(v < zero, -u, u)
        # marginals: LLR (0 for punctured bits, :382-383), then the variable's edges in edge order
        va = np.zeros((len(live), V), dtype=wt)
        va[:, :n] = L[live]
        for grp in st.rank_groups:
            va[:, st.var[grp]] = add(va[:, st.var[grp]], u[:, grp])
        # ---- pass 2 (:418-450)
        vae = va[:, st.var]
        nv = sub(vae, u)
        keep = ((nv < zero) == (v < zero)) | (v == zero)
        v = np.where(keep, nv, zero).astype(wt)
        am = mag(v)
        if is_float:
            am = np.where(np.isnan(am), pad, am)  # a NaN magnitude never passes the `<` of :430 / :433: it is +inf to the minima
        a = np.concatenate([am, np.broadcast_to(pad, (len(live), 1))], axis=1)[:, st.by_check]   # [f, C, maxdeg]
        a.sort(axis=2)
        min1 = np.minimum(a[:, :, 0], maxval).astype(wt)
        min2 = np.minimum(a[:, :, 1], maxval).astype(wt)
        neg_v = np.concatenate([v < zero, np.zeros((len(live), 1), dtype=bool)], axis=1)[:, st.by_check]
        sgn = np.logical_xor.reduce(neg_v, axis=2)
        neg_va = np.concatenate([vae < zero, np.zeros((len(live), 1), dtype=bool)], axis=1)[:, st.by_check]
        parity = np.logical_xor.reduce(neg_va, axis=2)
        done = ~parity.any(axis=1)                # all checks satisfied (:453)
        if done.any():
            fr = live[done]
            output[fr] = hard_pack(va[done])
            iters[fr] = it                        # 0-based index of the converging iteration (:462)
            success[fr] = 1
            stay = ~done
            live, v, min1, min2, sgn, va = live[stay], v[stay], min1[stay], min2[stay], sgn[stay], va[stay]
    if len(live):
        output[live] = hard_pack(va)              # failure: hard decision of the last marginals (:466-474)
    return output, iters, success
