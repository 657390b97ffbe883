/* ldpc_oracle.h -- CPU oracle for the labrador-ldpc min-sum hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (adamgreig/labrador-ldpc v1.2.1), used as the checker for the HIP
 * path and as the "port" CPU baseline in bench.py.  Nothing in the product
 * (labrador_ldpc_amd/, include/) may include, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity pinning: the reference is Rust and no rustc/cargo exists in the build
 * image, so oracle/_ref cannot be built ("unbuildable here").  The oracle is
 * pinned instead against every known-answer the reference's own tests hold for
 * this path (tests/golden/reference_kats.json, checked by
 * tests/test_oracle_kats.py):
 *   - edge order + count CRC-32 for all 9 codes      (src/codes/mod.rs:517-535)
 *   - encoder parity bytes for all 9 codes           (src/encoder.rs:361-527)
 *   - decode_ms i8 three-flip scenario, 9 codes      (src/decoder.rs:671-699)
 *   - decode_ms on clean punctured codewords         (src/decoder.rs:607-645)
 *   - hard_to_llrs / llrs_to_hard vectors            (src/decoder.rs:553-605)
 *   - working/output length tables                   (src/decoder.rs:531-551)
 *   - decode_bf three-flip scenario, decode_erasures  (src/decoder.rs:607-670, src/lib.rs:21-50)
 * Iteration counts, non-converged outputs and noisy soft inputs are NOT
 * asserted by any reference test: for those the oracle is pinned only by its
 * fidelity to src/decoder.rs:347-475 (stated in DESIGN.md).
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Code ids follow `enum LDPCCode` (src/codes/mod.rs:37-66): TC128=0 .. TM8192=8. */
#define ORACLE_NUM_CODES 9

/* Code parameters (src/codes/mod.rs:69-241, accessors :367-409). Return 0 for a bad id. */
size_t oracle_code_n(int code);
size_t oracle_code_k(int code);
size_t oracle_code_punctured_bits(int code);
size_t oracle_code_submatrix_size(int code);
size_t oracle_code_circulant_size(int code);
size_t oracle_code_paritycheck_sum(int code);

/* Buffer sizes (src/decoder.rs:93-116). */
size_t oracle_bf_working_len(int code);
size_t oracle_ms_working_len(int code);
size_t oracle_ms_working_u8_len(int code);
size_t oracle_output_len(int code);

/* Edge list in the exact order of `iter_paritychecks()` (src/codes/mod.rs:275-362,
 * :435-494).  Writes up to `cap` (check,var) pairs, returns the edge count E. */
size_t oracle_edges(int code, uint16_t *checks, uint16_t *vars, size_t cap);

/* CRC-32 over the edge stream exactly as test_iter_parity computes it
 * (src/codes/mod.rs:508-533). */
uint32_t oracle_edge_crc(int code);

/* decode_ms<T> (src/decoder.rs:347-475).  Returns 1 on success, 0 on failure,
 * -1 on a bad code id.  `iters_run` may be NULL (capi/src/lib.rs:91-93). */
int oracle_decode_ms_i8 (int code, const int8_t  *llrs, uint8_t *output, int8_t  *working,
                         uint8_t *working_u8, size_t maxiters, size_t *iters_run);
int oracle_decode_ms_i16(int code, const int16_t *llrs, uint8_t *output, int16_t *working,
                         uint8_t *working_u8, size_t maxiters, size_t *iters_run);
int oracle_decode_ms_i32(int code, const int32_t *llrs, uint8_t *output, int32_t *working,
                         uint8_t *working_u8, size_t maxiters, size_t *iters_run);
int oracle_decode_ms_f32(int code, const float   *llrs, uint8_t *output, float   *working,
                         uint8_t *working_u8, size_t maxiters, size_t *iters_run);
int oracle_decode_ms_f64(int code, const double  *llrs, uint8_t *output, double  *working,
                         uint8_t *working_u8, size_t maxiters, size_t *iters_run);

/* Batched driver used by tests and by bench.py's cpu_baseline leg: runs the
 * single-codeword decoder above on `batch` frames laid out [batch][n], one
 * thread per core with private working buffers (the spawn_broadcast pattern
 * of perftest/src/main.rs:39-45).  nthreads<=0 means all cores.
 * Returns the number of threads used, or -1 on error. */
int oracle_decode_ms_batch_i8 (int code, const int8_t  *llrs, uint8_t *output, uint32_t *iters,
                               uint8_t *success, size_t batch, size_t maxiters, int nthreads);
int oracle_decode_ms_batch_i16(int code, const int16_t *llrs, uint8_t *output, uint32_t *iters,
                               uint8_t *success, size_t batch, size_t maxiters, int nthreads);
int oracle_decode_ms_batch_i32(int code, const int32_t *llrs, uint8_t *output, uint32_t *iters,
                               uint8_t *success, size_t batch, size_t maxiters, int nthreads);
int oracle_decode_ms_batch_f32(int code, const float   *llrs, uint8_t *output, uint32_t *iters,
                               uint8_t *success, size_t batch, size_t maxiters, int nthreads);
int oracle_decode_ms_batch_f64(int code, const double  *llrs, uint8_t *output, uint32_t *iters,
                               uint8_t *success, size_t batch, size_t maxiters, int nthreads);

/* decode_bf (src/decoder.rs:243-301) and its erasure pre-pass decode_erasures (:144-223).
 * `working` has bf_working_len bytes.  Return 1 / 0 (success), -1 bad code. */
int oracle_decode_bf(int code, const uint8_t *input, uint8_t *output, uint8_t *working,
                     size_t maxiters, size_t *iters_run);
int oracle_decode_erasures(int code, uint8_t *codeword, uint8_t *working, size_t maxiters, size_t *iters_run);

/* hard_to_llrs / llrs_to_hard (src/decoder.rs:484-509). */
void oracle_hard_to_llrs_i8 (int code, const uint8_t *input, int8_t  *llrs);
void oracle_hard_to_llrs_i16(int code, const uint8_t *input, int16_t *llrs);
void oracle_hard_to_llrs_i32(int code, const uint8_t *input, int32_t *llrs);
void oracle_hard_to_llrs_f32(int code, const uint8_t *input, float   *llrs);
void oracle_hard_to_llrs_f64(int code, const uint8_t *input, double  *llrs);
void oracle_llrs_to_hard_i8 (int code, const int8_t  *llrs, uint8_t *output);
void oracle_llrs_to_hard_i16(int code, const int16_t *llrs, uint8_t *output);
void oracle_llrs_to_hard_i32(int code, const int32_t *llrs, uint8_t *output);
void oracle_llrs_to_hard_f32(int code, const float   *llrs, uint8_t *output);
void oracle_llrs_to_hard_f64(int code, const double  *llrs, uint8_t *output);

/* Systematic encoder with the behaviour of encode / copy_encode
 * (src/encoder.rs:293-315): the first k/8 bytes are data, the remaining
 * (n-k)/8 bytes are written with parity.  The generator is not carried as
 * constants: it is derived from H by GF(2) elimination (the systematic
 * generator of a full-rank H is unique), pinned by the 9 parity KATs.
 * Return 0 on success, -1 on a bad code / singular matrix. */
int oracle_encode(int code, uint8_t *codeword);
int oracle_copy_encode(int code, const uint8_t *data, uint8_t *codeword);

/* Syndrome weight of an (n+p)-bit word packed MSB-first: number of unsatisfied
 * checks of H.  0 <=> valid codeword.  Test helper (no reference counterpart). */
size_t oracle_syndrome_weight(int code, const uint8_t *bits_np);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_ORACLE_H */
