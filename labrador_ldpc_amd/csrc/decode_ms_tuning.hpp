// decode_ms_tuning.hpp -- the tuned settings of the min-sum kernels, in one place.
//
// A LIBRARY build uses exactly the values below: a stray -DLDPC_... in the build's flags stops it here instead of shipping a mistuned
// decoder.  Same-process A/B experiments on these settings, and the timing diagnostics that leave pieces of the decoder out, live
// outside the library: tools/kbench/ applies its overlay (tools/kbench/diag_overlay.patch) to a COPY of these sources and builds its
// own binaries from that (tools/kb_build.sh); nothing of it is visible to the library's translation units.
#pragma once

#ifndef LDPC_TUNING_OVERRIDES_ALLOWED
#if defined(LDPC_QUARTER_SPECIALISE) || defined(LDPC_NOCAP) || defined(LDPC_LOCAL_IN_VAR) || defined(LDPC_PRIO) || defined(LDPC_PRIO_ROWS) || \
    defined(LDPC_PRIO_ROWS_LEAN) || defined(LDPC_PRIO_VAR) || defined(LDPC_TM2048_WAVES) || defined(LDPC_PAIR_LOCAL_IN_VAR) || defined(LDPC_PAIR_NOCAP) || \
    defined(LDPC_PAIR_ODD_B64) || defined(LDPC_PRIO_ROWS_PAIR) || defined(LDPC_SELFCORR_CARRY) || defined(LDPC_WAVE_VERDICT) || defined(LDPC_WG_VERDICT) || \
    defined(LDPC_PEEL_FIRST) || defined(LDPC_PAIR_PEEL_FIRST) || defined(LDPC_PAIR_FETCH_EARLY) || defined(LDPC_PAIR_SELFCORR_CARRY) || \
    defined(LDPC_SELFCORR_MED3) || defined(LDPC_PAIR_SELFCORR_MED3) || defined(LDPC_LEAN_VERDICT) || defined(LDPC_LEAN_CH)
#error "the LDPC_* tuning switches are fixed in a library build (kernel experiments: tools/kbench/)"
#endif
#endif

// ---- decode_ms_kernel.hpp ---------------------------------------------------------------------------------
// Quarter-specialised bodies (one copy per starting quarter, every pi_k constant a literal): 2 = for
// workgroups spanning two or four quarters (+4.6 % on TM8192), 1 = two only, 0 = off.
#ifndef LDPC_QUARTER_SPECIALISE
#define LDPC_QUARTER_SPECIALISE 2
#endif
// Self-correction select through the borrow of an integer subtraction (I class) instead of a float compare (C class):
// -1 = per code (selfcorr_carry_default()), 0 = off, 1 = i8/i16, 2 = f32 too.
#ifndef LDPC_SELFCORR_CARRY
#define LDPC_SELFCORR_CARRY -1
#endif
// Self-correction as a clamp, v = med3(nv, 0, nv + old * big) (Ops<float>::clamp_to_side): -1 = per kernel
// (selfcorr_med3()), 0 = off, 2 = v_fma form, 3 = v_mul_legacy form, 5 = three full-rate operations, no median.
#ifndef LDPC_SELFCORR_MED3
#define LDPC_SELFCORR_MED3 -1
#endif
#ifndef LDPC_PAIR_SELFCORR_MED3
#define LDPC_PAIR_SELFCORR_MED3 -1
#endif
// pair kernel: where the next codeword's LLR loads are issued
#ifndef LDPC_PAIR_FETCH_EARLY
#define LDPC_PAIR_FETCH_EARLY -1
#endif
// iteration 0 as a pass of its own on the pair kernel (u = v = 0 folded, no slot zeroing): TM8192 f32 7.51 -> 7.70,
// i8 6.81 -> 7.02 M codewords/s.  (Round 1's version peeled only the variable phase and lost: 7.42 -> 6.94.)
#ifndef LDPC_PAIR_PEEL_FIRST
#define LDPC_PAIR_PEEL_FIRST 1
#endif
// the same select on the pair kernel (measured slower there: 7.25-7.40 against 7.50)
#ifndef LDPC_PAIR_SELFCORR_CARRY
#define LDPC_PAIR_SELFCORR_CARRY 0
#endif
// Codewords inside one wave (TC codes): barrier-free verdict at the top of the check phase, skipping the rest of the
// phase on success (WAVE_VERDICT in the kernel body).
#ifndef LDPC_WAVE_VERDICT
#define LDPC_WAVE_VERDICT 1
#endif
// iteration 0 as a pass of its own (u = v = 0 folded): -1 = per kernel (peel_first_default()), 0 / 1 = force
#ifndef LDPC_PEEL_FIRST
#define LDPC_PEEL_FIRST -1
#endif
// the same for multi-wave codewords through a third barrier per iteration: -1 = per code and type (WG_VERDICT_SET)
#ifndef LDPC_WG_VERDICT
#define LDPC_WG_VERDICT -1
#endif
// register-lean kernels: in-phase verdict through a third barrier, skipping the minima / next-u half of the last check phase
// (LEAN_VERDICT in the kernel body): -1 = per kernel (lean_verdict_default())
#ifndef LDPC_LEAN_VERDICT
#define LDPC_LEAN_VERDICT -1
#endif
// f32: clamp-free check phase for codewords whose LLRs are bounded (NOCAP_POSSIBLE in the kernel body).
#ifndef LDPC_NOCAP
#define LDPC_NOCAP 1
#endif
// Local-edge updates done at the end of the variable phase: -1 = per kernel (local_in_var_default()).
#ifndef LDPC_LOCAL_IN_VAR
#define LDPC_LOCAL_IN_VAR -1
#endif
// Progress-based wave priority.  VALU issue is arbitrated by priority, then age, so the oldest wave
// of a SIMD runs ahead and the youngest arrives last at every barrier, the last stretch of each
// phase with the SIMD half empty.  Lowering a wave's priority as it advances through a phase
// lets the laggards catch up.  The schedule that measured best keeps priority 3 through the edge
// updates and steps down over the last check rows (LDPC_PRIO_ROWS; a sweep of a dozen schedules spans
// 6.06-6.40 M codewords/s on TM8192, the inverted one 5.54): TM8192 5.61 -> 6.40, TM6144 9.64 ->
// 10.83, TM2048 46.9 -> 48.1, TM5120 13.76 -> 14.02 M codewords/s.  Only for codewords of 8 or more waves
// (PRIO_WAVES in the kernel): with one or two waves per codeword it costs (TC512 -7 %, TM1280 -1 %).
// 0 = off, 1 = with a scheduling barrier at each step, 2 = plain.
#ifndef LDPC_PRIO
#define LDPC_PRIO 2
#endif
#ifndef LDPC_PRIO_ROWS
#define LDPC_PRIO_ROWS {2, 2, 1, 1, 1, 0}        // priority during the last six (index, check row) steps of the check phase
#endif
// edges per request chunk of the register-lean check phase (check_phase_lean)
#ifndef LDPC_LEAN_CH
#define LDPC_LEAN_CH 6
#endif
#ifndef LDPC_PRIO_ROWS_LEAN
#define LDPC_PRIO_ROWS_LEAN {3, 3, 3, 2, 1, 0}   // the same for the register-lean check phase
#endif
#ifndef LDPC_PRIO_VAR
#define LDPC_PRIO_VAR 2                          // priority of the first half of the (short) variable phase
#endif
// Waves per SIMD the TM2048 kernels' register allocation must leave room for (min_waves_per_simd()).
#ifndef LDPC_TM2048_WAVES
#define LDPC_TM2048_WAVES 6
#endif

// ---- decode_ms_pair.hpp (each measured with tools/kbench.hip -DKPAIR=1 on TM8192) ---------------------------
// How many of the thread's local-edge updates (of 14 on TM8192) are done at the end of the variable phase
// (LDS-bound: the VALU idles there) instead of at the start of the check phase (VALU-bound), where the
// rest still covers the latency of the marginal reads.  f32 0/3/5/7/9/14 -> 7.04 / 7.07 / 7.24 / 7.41 /
// 7.32 / 7.26 M codewords/s in round 1; re-swept after the multiply form of the self-correction test
// (round 2): 5/6/7/8/9/11 -> 7.23 / 7.30 / 7.40 / 7.49 / 7.41 / 7.14; i8 3/4/5/6/7 -> 6.70 / 6.79 / 6.75 / 6.76 /
// 6.73.  Round 3, with the clamp form of the self-correction (cheaper updates): f32 6/7/8/9/10/12 -> 8.00 / 8.03 / 8.30 / 8.28 /
// 8.33 / 8.15; i8 2/4/6/8 -> 7.43 / 7.45 / 7.35 / 7.55.  -1 = per type (10 for f32, 8 else).
#ifndef LDPC_PAIR_LOCAL_IN_VAR
#define LDPC_PAIR_LOCAL_IN_VAR -1
#endif
// f32: run the check phase without the FLT_MAX clamp of the exclusive minimum when no LLR of the codeword
// exceeds the launch's nocap_limit in magnitude (see begin_codeword): 6.84 -> 7.05.
#ifndef LDPC_PAIR_NOCAP
#define LDPC_PAIR_NOCAP 1
#endif
// Odd rotations read their two marginals as halves of two aligned 64-bit pairs: -1 = the default (on: i8/i16 8.0 -> 9.0 in
// round 1; f32 7.05 -> 6.90 then, 8.33 -> 8.44 with round 3's check phase), 0 / 1 = force.
#ifndef LDPC_PAIR_ODD_B64
#define LDPC_PAIR_ODD_B64 -1
#endif

// Wave priority over the six (check row, index) steps of the check phase; 3 before them.  A dozen
// alternatives, also per quarter, measured 6.4-6.75 against 6.75 for this one.
#ifndef LDPC_PRIO_ROWS_PAIR
#define LDPC_PRIO_ROWS_PAIR {3, 3, 2, 2, 1, 0}
#endif
