/*
 * include/celerite_hip_debug.h -- diagnostics, probes and test hooks of libcelerite_hip.so.  NOT part of the drop-in
 * boundary (include/celerite_hip.h): nothing a caller of the reference's CholeskySolver / GP path needs, no stability
 * promise.  Used by tests/, tools/ and bench.py (the measured fp64 rate of the box, the fp32-state probe of BASELINE
 * configs[4]'s "fp32 vs fp64 tolerance", the scanned start states, the element-composition check, the CU census of
 * the materialising pipeline's streams).
 */
#ifndef CELERITE_HIP_DEBUG_H
#define CELERITE_HIP_DEBUG_H

#include "celerite_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A measurement for roofline figures (not a product path): the fp64 FMA rate the vector ALUs of the current device
 * sustain with `waves_per_simd` waves per SIMD issuing independent v_fma_f64 back to back (64 x iters each), the shader
 * clock during that load (median over the waves: s_memtime against the 100 MHz s_memrealtime) and the cycles a SIMD
 * spends per FMA it issues.  MI355X: ~56 TFLOP/s at ~1.9 GHz and ~4.45 cycles with two waves per SIMD -- the datasheet's
 * 78.6 TFLOP/s assumes 4 cycles at 2.4 GHz.  Any pointer may be NULL. */
int clr_device_measure_fp64(int waves_per_simd, int iters, double* tflops, double* clock_mhz, double* cycles_per_fma);

/* Which route the last clr_solver_compute took through the chunked flow (widths 1..64 without general terms, more than
 * one chunk; otherwise level = -1): level 0 / 1 the chunked replay from the scanned start states stands, 2 the sequential
 * recurrence; residual = the largest entry of the chunks' record -- after a replay that met the scanned start states its
 * relative end-state mismatch, after the output check (BatchParams::head_check; a state mismatch above the certificate's
 * bound) the largest mismatch of what the two replays wrote. */
int clr_solver_debug_route(const clr_solver* s, int* level, int* nchunk, double* residual);

/* Diagnostics (tests): the chunk start states of the last evaluation, [B][nchunk][J(J+1)/2 + J] (packed upper
 * triangle of P, then f) ... */
int clr_batch_debug_get_starts(clr_batch* h, double* starts);

/* ... and the cooperative composition kernel against the single-lane host-checked form on the last evaluation's
 * chunk elements in groups of `group`: the largest difference relative to the largest entry of the same block
 * (A, b, C, eta, Jm) of the same composed element, and the largest magnitude seen. */
int clr_batch_debug_compose_check(clr_batch* h, int group, double* max_rel_diff, double* max_abs_value);

/* Diagnostic: on how many distinct compute units of each of the 8 XCDs a grid launched on the plan's stream (which =
 * 0), on the pipeline's first summarize stream (1) or on its replay stream (2) runs. */
int clr_batch_debug_cu_census(clr_batch* h, int which, int* cus_per_xcc /* [8] */);

/* A measurement, not a product path (BASELINE config 5 asks for the fp32-vs-fp64 tolerance of the wide
 * recurrence): the sequential sweep of a width 9..32 plan with the state and every per-step operation in
 * float (features in fp64, rounded; log det and the quadratic form accumulated in fp64 from the float
 * pivots).  Returns per-problem log det / quadratic form and the kernel's time for the whole batch. */
int clr_batch_fp32_probe(clr_batch* h, double* logdet, double* quad, double* ms);

#ifdef __cplusplus
}
#endif
#endif
