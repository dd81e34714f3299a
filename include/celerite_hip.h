/*
 * include/celerite_hip.h -- C ABI of libcelerite_hip.so, the MI355X (gfx950)
 * implementation of celerite's semiseparable-Cholesky hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one member of
 * the reference's C++ solver class as it is bound by the reference's pybind11
 * translation unit (celerite/solver.cpp).  The reference-side binding a
 * maintainer would add is shown in INTEGRATION.md.  Conventions:
 *
 *   - extern "C", plain pointers + sizes, no C++ types, no exceptions: every
 *     call returns a clr_status (the pybind11 layer maps CLR_NOT_POSITIVE_DEFINITE
 *     -> celerite.solver.LinAlgError and the other non-zero codes ->
 *     RuntimeError with the reference's what() strings, exceptions.h:14-36).
 *   - all pointers are caller-owned HOST memory unless a name says `_dev`;
 *     inputs are const and are not retained past the call (the reference's
 *     pybind11/Eigen casters copy every argument, solver.cpp:467-483).
 *   - a handle is thread-compatible (one thread at a time), like the reference
 *     object; distinct handles may be used concurrently.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry returns CLR_NO_DEVICE.
 *   - fp64 throughout.  Matrices of the factor use the reference's storage
 *     (Eigen column-major J x N: element (j, n) at [j + J*n], cholesky.h:703-706).
 *     U and V (general terms) are ROW-major [J_general][N], as the NumPy arrays
 *     arrive at the Python boundary.
 */
#ifndef CELERITE_HIP_H
#define CELERITE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum clr_status {
  CLR_OK = 0,
  CLR_DIMENSION_MISMATCH = 1,    /* celerite::dimension_mismatch  exceptions.h:20-24 */
  CLR_NOT_POSITIVE_DEFINITE = 2, /* celerite::linalg_exception    exceptions.h:32-36 */
  CLR_NOT_COMPUTED = 3,          /* celerite::compute_exception   exceptions.h:14-18 */
  CLR_NO_DEVICE = 4,             /* no gfx950 GPU visible / HIP runtime unusable    */
  CLR_HIP_ERROR = 5,             /* a HIP call failed; see clr_last_error()          */
  CLR_INVALID_ARGUMENT = 6,
  CLR_UNSUPPORTED = 7,           /* e.g. width above CLR_MAX_WIDTH                   */
  CLR_CARMA_INSTABILITY = 8      /* celerite::carma_exception     exceptions.h:8-12  */
} clr_status;

/* Widest semiseparable rank J = J_real + 2 J_comp + J_general of the batched plans (fused log-likelihood; the scans stop
 * at 64) and of the one-workgroup kernels of the object API. */
#define CLR_MAX_WIDTH 128
/* The object API -- CholeskySolver.compute / log_determinant / dot_solve / solve / dot_L / dot / predict and the pickled
 * state -- takes ANY width up to this one (round 6; the reference's dynamic-width arm, cholesky.h:203, has no limit and
 * its benchmark goes to 512): compute above 64 keeps S in the registers of 1 .. 64 workgroups (csrc/rows_kernels.hip),
 * dot_L / dot are diagonal scans with one thread per row, solve / dot_solve above 64 chunked affine scans with
 * J x J chunk maps built once per factor (csrc/bigsweep_kernels.hip; short series: one workgroup per right-hand side);
 * grad_log_likelihood above width 64: one workgroup per partial, S and its tangent in an HBM / L2 workspace
 * (csrc/grad_any_kernels.hip; sequential in n). */
#define CLR_MAX_WIDTH_ANY 1024
#define CLR_CARMA_MAX_ORDER 32 /* autoregressive order p of clr_carma (state p, covariance p x p in LDS) */

/* ---- library / device ------------------------------------------------------ */

/* CELERITE_VERSION_STRING, cpp/include/celerite/version.h:4-13 ("0.3.0");
 * bound as solver.get_library_version (solver.cpp:76). */
const char* clr_version(void);
/* Message of the last non-OK status on the calling thread. */
const char* clr_last_error(void);
/* what() string of the reference exception a status stands for. */
const char* clr_status_string(int status);
/* Number of visible gfx950 devices (0 without a GPU; never fails). */
int clr_device_count(void);
/* Device used by handles created afterwards on this thread (default 0). */
int clr_set_device(int device);
int clr_get_device(int* device);
/* hipDeviceSynchronize on the current device. */
int clr_device_synchronize(void);
/* Name + CU count of the current device (for logs). */
int clr_device_info(char* name, size_t name_len, int* compute_units, size_t* hbm_bytes);
/* Free / total HBM of the current device right now (hipMemGetInfo): for sizing a batch to the 288 GB of a
 * GPU -- a plan of B problems x N samples x width J holds about 8 B N (3 + [3 J + 1 when materialising]) bytes
 * of series and factor plus 8 B nchunk (J^2 + J (J + 1) + 4 J) of scan workspace. */
int clr_device_memory(size_t* free_bytes, size_t* total_bytes);

/* Tuning and cross-check switches of the library, ONE table per process (round 5 read 19 environment variables
 * directly: a stray variable silently changed kernel selection).  `value` NULL removes the option.  Environment variables
 * of the same names are honoured only when the process was started with CLR_ALLOW_ENV=1.  The keys (all "CLR_..."):
 *   CLR_GRAD_SEQUENTIAL      the sequential tangent kernel for every gradient (cross-checks); with CLR_GRAD_ANY_WIDTH the
 *                            workgroup-per-partial kernel of the widths above 64 at every width
 *   CLR_GRAD_REBUILD_SPAN    reverse-mode gradient: distance of the stored states
 *   CLR_NO_SMALL_SOLVER      CholeskySolver: never the one-workgroup kernel of short narrow problems
 *   CLR_WIDE_WALK            widths 17..32: prefix + corrections as one walk per problem (cross-check of the two-kernel path)
 *   CLR_WIDE_PREFIX_WALK     CholeskySolver at widths 9..32: the sequential walk instead of the parallel prefix
 *   CLR_OUTPUT_CHECK_CAP, CLR_OUTPUT_CHECK_TOL   materialising replays at widths 9..64 whose end states miss the scanned start
 *                            states by more than the certificate's bound but less than CAP (default 1e-6; 0: off) are
 *                            settled by comparing what a second replay writes (within TOL, default 2e-11) instead of
 *                            going to the sequential recurrence; CLR_SOLVER_CERT_RESID: that bound for CholeskySolver (1e-11);
 *                            CLR_OUTPUT_CHECK_ATTEMPTS: replays before giving up (4)
 *   CLR_NO_ROWS_KERNEL       CholeskySolver above width 64 / general terms above 32: the one-workgroup kernels (S in LDS / L2)
 *                            instead of the row-distributed one (cross-checks); CLR_ROWS_NO_BLOCKS: its row-per-lane-group layout
 *   CLR_NO_BIG_SWEEP         dot_solve / solve above width 64: the sequential sweeps instead of the chunked affine scans
 *   CLR_WIDE_NO_PAIRED, CLR_WIDE64_ONE_WAVE, CLR_WIDE_LAZY_BOUND, CLR_WIDE_FIRST_RATIO, CLR_WIDE_FIRST_RATIO64,
 *   CLR_WIDE_SCAN_CAP, CLR_SOLVER_WIDE_CHUNKS, CLR_PREDICT_CHUNKS, CLR_WSWEEP_RUN, CLR_WSWEEP_CHUNKS   (tuning runs, tools/)
 * clr_get_option: the value in force (NULL: not set); the pointer is valid until the calling thread's next call. */
int clr_set_option(const char* key, const char* value);
const char* clr_get_option(const char* key);

/* ---- single-problem solver: celerite::solver::CholeskySolver<double> ----------
 * (cpp/include/celerite/solver/cholesky.h, solver.h), the object behind
 * celerite.solver.CholeskySolver (solver.cpp:241-244).  The factor
 * (phi, u, W, D) stays resident in HBM between calls. */

typedef struct clr_solver clr_solver;

clr_solver* clr_solver_create(void);       /* CholeskySolver()   solver.cpp:244 */
void clr_solver_destroy(clr_solver* s);

/* CholeskySolver::compute, cholesky.h:41-210 (bound at solver.cpp:467-483).
 * Each array carries its own length so the reference's dimension checks
 * (cholesky.h:59-69) are made here.  n_A == 0 means "no general terms".
 * On CLR_NOT_POSITIVE_DEFINITE (some D_n < 0, n >= 1: cholesky.h:176) or
 * CLR_DIMENSION_MISMATCH the handle is left "not computed" (cholesky.h:57). */
int clr_solver_compute(clr_solver* s, double jitter,
                       int n_a_real, const double* a_real,
                       int n_c_real, const double* c_real,
                       int n_a_comp, const double* a_comp,
                       int n_b_comp, const double* b_comp,
                       int n_c_comp, const double* c_comp,
                       int n_d_comp, const double* d_comp,
                       int n_A, const double* A,
                       int U_rows, int U_cols, const double* U,
                       int V_rows, int V_cols, const double* V,
                       int n_x, const double* x,
                       int n_diag, const double* diag);

/* Optional, not in the reference: announce the vector the caller is about to hand to dot_solve (GP.log_likelihood
 * knows y before it factorises, celerite.py:180-215).  The NEXT clr_solver_compute then folds b^T K^-1 b into its own
 * pass over the series (widths 1..64 without general terms) and clr_solver_dot_solve returns that value when it is
 * called with the very same vector (compared byte for byte); any other vector takes the ordinary sweep.  One shot:
 * the hint is consumed by the next clr_solver_compute whatever its outcome (a hint of another length is dropped
 * there); n_b = 0 withdraws a hint that will not be followed by a compute. */
int clr_solver_hint_rhs(clr_solver* s, int n_b, const double* b);

/* Solver::computed / log_determinant, solver.h:74-81 (solver.cpp:620-634). */
int clr_solver_computed(const clr_solver* s);
int clr_solver_log_determinant(const clr_solver* s, double* out);

/* CholeskySolver::dot_solve, cholesky.h:326-401 (solver.cpp:527-529):
 * out = b^T K^-1 b. */
int clr_solver_dot_solve(const clr_solver* s, int n_b, const double* b, double* out);

/* CholeskySolver::solve, cholesky.h:218-318 (solver.cpp:507-509).
 * b, x: COLUMN-major (N, nrhs): right-hand side k is b + k*N. */
int clr_solver_solve(const clr_solver* s, int b_rows, int nrhs, const double* b, double* x);

/* CholeskySolver::dot_L, cholesky.h:409-431 (solver.cpp:547-549): y = L z with
 * K = L L^T.  Column-major (N, nrhs). */
int clr_solver_dot_L(const clr_solver* s, int z_rows, int nrhs, const double* z, double* y);

/* CholeskySolver::dot, cholesky.h:444-590 (solver.cpp:567-581): y = K z without
 * factorising (stateless; the handle only supplies the device).  z, y:
 * column-major (N, nrhs). */
int clr_solver_dot(clr_solver* s, double jitter,
                   int n_a_real, const double* a_real,
                   int n_c_real, const double* c_real,
                   int n_a_comp, const double* a_comp,
                   int n_b_comp, const double* b_comp,
                   int n_c_comp, const double* c_comp,
                   int n_d_comp, const double* d_comp,
                   int n_A, const double* A,
                   int U_rows, int U_cols, const double* U,
                   int V_rows, int V_cols, const double* V,
                   int n_x, const double* x,
                   int z_rows, int nrhs, const double* z, double* y);

/* CholeskySolver::predict, cholesky.h:599-698 (solver.cpp:611-615): conditional
 * mean at the M sorted-or-not coordinates xs given observations y. */
int clr_solver_predict(const clr_solver* s, int n_y, const double* y,
                       int M, const double* xs, double* pred);

/* grad_log_likelihood, celerite/solver.cpp:347-463 (the reference runs compute +
 * dot_solve on forward-mode dual numbers; the handle's factorisation is neither used
 * nor changed, as in the reference where the bound object is ignored).  Same argument
 * conventions as clr_solver_compute plus the observations y.
 *   *value = -(y^T K^-1 y + log det K + pi log N) / 2   (the reference's constant,
 *            solver.cpp:415 -- not N log 2 pi);
 *   grad[0] = d/d jitter (0 when jitter <= DBL_EPSILON, solver.cpp:379-389,419-426),
 *   then d/d a_real, c_real, a_comp, b_comp, c_comp, d_comp: n_grad must be
 *   1 + 2 n_a_real + 4 n_a_comp.
 * Returns CLR_NOT_POSITIVE_DEFINITE where the reference throws linalg_exception.
 * Any total width up to CLR_MAX_WIDTH_ANY: widths 1..8 from N = 1024 and widths 9..64 from N = 4096 run parallel in n on a
 * one-problem plan (general terms: up to a total width of 32); otherwise one wave per partial, sequential in n, up to
 * width 64 (csrc/grad_kernels.hip) and one workgroup per partial above (round 6: csrc/grad_any_kernels.hip). */
int clr_solver_grad_log_likelihood(clr_solver* s, double jitter,
                                   int n_a_real, const double* a_real,
                                   int n_c_real, const double* c_real,
                                   int n_a_comp, const double* a_comp,
                                   int n_b_comp, const double* b_comp,
                                   int n_c_comp, const double* c_comp,
                                   int n_d_comp, const double* d_comp,
                                   int n_A, const double* A,
                                   int U_rows, int U_cols, const double* U,
                                   int V_rows, int V_cols, const double* V,
                                   int n_x, const double* x,
                                   int n_y, const double* y,
                                   int n_diag, const double* diag,
                                   double* value, int n_grad, double* grad);

/* PicklableCholeskySolver::serialize / deserialize, solver.cpp:36-58 (bound as
 * __getstate__/__setstate__, solver.cpp:644-663).  get_dims first, then
 * get_state into caller buffers of J*(N-1), J*(N-1), J*N and N doubles
 * (reference storage order). */
int clr_solver_get_dims(const clr_solver* s, int* computed, int* N, int* J, double* log_det);
int clr_solver_get_state(const clr_solver* s, double* phi, double* u, double* W, double* D);
int clr_solver_set_state(clr_solver* s, int computed, int N, int J, double log_det,
                         const double* phi, const double* u, const double* W, const double* D);

/* ---- batched log-likelihood (new; the data-parallel axis) ---------------------
 * B independent problems = (time series x hyper-parameter draw) pairs that
 * share N, J_real and J_comp.  One call evaluates, for every problem p,
 *     loglike[p] = -0.5 (y^T K_p^-1 y + log det K_p + N log 2 pi)
 * i.e. exactly GP.compute + GP.log_likelihood (celerite/celerite.py:103-219)
 * with the -inf rules of celerite.py:205-218 applied (quiet=True semantics:
 * status[p] = CLR_NOT_POSITIVE_DEFINITE and loglike[p] = -inf instead of an
 * exception; a bad problem never disturbs its neighbours).
 *
 * Coefficients are [B][J_real] / [B][J_comp] row-major, jitter is [B].
 * A series array is [B][N] with *_stride = N, or one shared [N] series with
 * *_stride = 0 (the "one light curve, B posterior draws" case). */

typedef struct clr_batch clr_batch;

/* Plans device buffers + workspace for (B, N, J_real, J_comp) on `device`.
 * General terms: clr_batch_set_general.  Width W = J_real + 2 J_comp:
 *   1..8   chunked scan over n, one lane per (problem, chunk)  (the headline path);
 *   9..64  one wave per (problem, chunk), S distributed over the lanes (BASELINE
 *          config 5: 16 complex terms); fused log-likelihood only.  One chunk = the
 *          reference recurrence itself; the series is cut into chunks (two-pass
 *          scan) when B alone leaves SIMDs idle (clr_batch_set_chunks(h, 0) picks
 *          2048 / B, at most 16, at widths <= 32; 1024 / B, at most 16, at widths 33..64 -- one
 *          wave per SIMD there, the chunks chained by one walk per problem: csrc/wide64_kernels.hip).
 *          Materialising runs write the reference's storage directly
 *          (every chunk replayed from its scanned start state and checked, as CholeskySolver.compute
 *          does); layouts do not apply;
 *   65..128 (round 5) the any-width sequential recurrence, one workgroup per problem with S in LDS (the kernel of
 *          plans with general terms): fused log-likelihood only, sequential in n;
 *   else   CLR_UNSUPPORTED (the reference's dynamic-width arm, cholesky.h:203, takes any J). */
clr_batch* clr_batch_create(int B, int N, int J_real, int J_comp, int device);
void clr_batch_destroy(clr_batch* h);

/* Host -> HBM. */
int clr_batch_set_series(clr_batch* h,
                         const double* t, long t_stride,
                         const double* diag, long diag_stride,
                         const double* y, long y_stride);
/* Short narrow problems (widths 1..4, 512 <= N <= 32768; BASELINE configs[1]: 256 x 1e4 x width 4): the whole fused
 * evaluation in ONE launch, one workgroup per problem -- chunk elements composed by a Kogge-Stone scan in LDS, the
 * reference recurrence per chunk from the scanned states, every chunk boundary checked; a problem that fails the check
 * (or has a flagged pivot) goes through the scan pipeline before results are handed out.  mode: -1 automatic (up to
 * 1024 problems), 0 off, 1 whenever supported.  clr_batch_get_small_mode: whether the next evaluation takes it. */
int clr_batch_set_small_mode(clr_batch* h, int mode);
int clr_batch_get_small_mode(const clr_batch* h, int* active);
/* The smallest step t[n + 1] - t[n] over the plan's series, found by the device-side scan of clr_batch_set_series
 * (negative: some series is not sorted -- GP.compute's check, celerite.py:126-129, without a host pass over t -- and
 * that holds whatever else the batch contains: the minimum is taken over the finite steps, as np.diff(t) < 0 would;
 * NaN: no negative step, but a NaN time somewhere).  clr_batch_clear_series drops the series (a front end that
 * rejects unsorted input calls it).
 * clr_batch_set_series replaces the plan's series IN PLACE: from the moment it starts copying, the previous series is
 * gone -- a call that fails midway, or whose input the front end then rejects, leaves the plan WITHOUT a series
 * (set one again before the next evaluation), never with a half-overwritten one. */
int clr_batch_get_series_order(const clr_batch* h, double* dtmin);
int clr_batch_clear_series(clr_batch* h);
/* (jitter may be NULL: no jitter) */
int clr_batch_set_coefficients(clr_batch* h, const double* jitter,
                               const double* a_real, const double* c_real,
                               const double* a_comp, const double* b_comp,
                               const double* c_comp, const double* d_comp);

/* General semiseparable terms for the whole batch (cholesky.h:65-72,148-152: A added to the diagonal, rows U, V
 * appended to U~, V~ with phi = 1): A [N], U and V row-major [J_general][N] per problem, *_stride doubles between
 * problems (N resp. J_general * N) or 0 for one block shared by all problems; J_general = 0 removes them.  A plan
 * with general terms evaluates, up to a total width J_real + 2 J_comp + J_general of 64, on the wave-per-(problem,
 * chunk) kernels of the widths 9..64 with the general rows as a third row class (per-sample features fetched a few
 * steps ahead; chunked scan up to total width 64, as clr_batch_set_chunks(h, 0) would pick for a wide plan), above
 * that -- or after clr_batch_set_general_route(h, 1) -- through the any-width sequential kernel (one workgroup per
 * problem, the reference's step order) up to CLR_MAX_WIDTH; fused log-likelihood only (materialising runs:
 * CholeskySolver). */
int clr_batch_set_general(clr_batch* h, int J_general, const double* A, long A_stride, const double* U, long U_stride,
                          const double* V, long V_stride);
int clr_batch_set_general_route(clr_batch* h, int route);

/* Tuning: how the kernels read the series.
 *   2 (default) staged: each wave loads the row-major arrays in coalesced tiles of
 *       8 steps x 64 chunks and transposes them through LDS; no extra pass or copy.
 *   1 interleaved: enqueue first builds a chunk-interleaved copy ([problem][i][chunk])
 *       with a tiled-transpose kernel whenever the series or the chunking changed;
 *       marginally faster per evaluation when the SAME series are evaluated many
 *       times, but costs a 0.87 ms pass (B=1024, N=1e5) whenever they change.
 *   0 row-major direct: every lane streams its own run (slow; kept for A/B). */
int clr_batch_set_layout(clr_batch* h, int layout);
/* The kernels evaluate sin/cos(d_comp * t) with a 25-instruction FMA Cody-Waite
 * routine (abs. error < 1 ulp(1)) when the host-side check max|d_comp| * max|t| <
 * 1e9 holds, and with the library (ocml) sincos otherwise.  force != 0 selects the
 * library routine unconditionally (for A/B measurements). */
int clr_batch_set_library_trig(clr_batch* h, int force);
/* summarize kernel of widths 7 and 8 (csrc/clr_split_kernels.h):
 *   0  the single-wave kernel (one wave per SIMD, part of the state in AGPRs);
 *   1  two roles on two waves per SIMD: a "trajectory" wave (C, b) and a "riders" wave (A, eta in
 *      registers, Jm in LDS) linked through LDS; reads a chunk-interleaved copy of the series made
 *      once per clr_batch_set_series (layout 1 below is implied);
 *   2  the same with the decay factored out of the state ("lazy": one FMA per state entry and step
 *      instead of FMA + MUL, renormalised every 64 steps) when the series is densely sampled
 *      (max c * max dx < 2^-7 and max d * max dx < 2^-5: the phases then advance by small-angle
 *      rotations, re-anchored with the full sincos every 64 steps), else as 1;
 *  -1  (default) 2 at widths 7 and 8 on a dense series; otherwise 1 at width 7 and at width 8 with at least two
 *      complex terms, else 0.
 * Wide plans (widths 9..64: csrc/wide_kernels.hip) know two flavours: 0 plain, 2 (and -1 where eligible) lazy.  At widths
 * 17..64 the lazy flavour is eligible whenever max c * max dx < 2 -- a lane whose interval is too long for the small-step
 * series sends its wave through the full sincos / exp for that batch -- at widths 9..16 under the dense-series rule above. */
int clr_batch_set_summarize_mode(clr_batch* h, int mode);
/* Series that FORGET their past (the decay between samples is not small: e.g. the paper's accuracy family,
 * paper/figures/error/error.py:24-25, or most real light curves) do not need the scan: every chunk runs the
 * reference recurrence itself from the zero state `K` samples before its first sample, and the state it has reached
 * at its first sample is checked against the state the previous chunk reaches there (relative mismatch <= the
 * max_residual of clr_batch_set_certificate); the first chunk starts from the true zero state, so agreement at every
 * boundary certifies all of them.  A problem with a mismatch, a flagged pivot or no usable K goes through the scan
 * pipeline when the results are fetched.  mode -1 (default): per problem, the smallest K of 8, 12, 16, ... 128 with
 * exp(-c_min x (time the K samples before any chunk boundary span)) <= exp(-32), used when at least half of the batch
 * has one; 0: off; 1: every problem with `forced_warmup` steps (tests: the check then decides).  Widths 1..8, fused
 * log-likelihood only (materialising / forced-exact runs take the scan). */
int clr_batch_set_warm_start(clr_batch* h, int mode, int forced_warmup);
/* What the current (series, coefficients) pair runs and how the last evaluation went: active, the warm path's chunk
 * count and length, smallest / largest warm-up in the batch, problems it settled / left to the scan (after the last
 * clr_batch_get_results).  Any pointer may be NULL. */
int clr_batch_get_warm_start(const clr_batch* h, int* active, int* nchunk, int* chunk_len, int* warmup_min,
                             int* warmup_max, int* settled, int* fallbacks);
/* Where the replay pass (materialising / forced-exact runs) reads the series when the summarize kernel reads the
 * chunk-interleaved copy: 0 that copy (default, also -1: 4.41 ms = 65 % of HBM for the materialising replay at
 * B = 1024, N = 1e5, width 8), 1 the row-major arrays through LDS-staged tiles (4.77 ms = 60 %;
 * profiles/r03a_prefix_ab.txt). */
int clr_batch_set_replay_source(clr_batch* h, int source);
/* Which one the next evaluation will run (0, 1 or 2 as above), given the series and coefficients set. */
int clr_batch_get_summarize_kernel(const clr_batch* h, int* kind);
/* Prefix phase of the scan (widths 1..8): how the chunk elements are turned into chunk start states.
 *   2 (default) multi-level: groups of consecutive elements are composed in parallel (element o element, the
 *       associative operator of the scan; csrc/clr_prefix_kernels.h), the few composed elements are walked, and the
 *       start states fan out group by group -- the dependent chain is ~2.2 (g - 1) per level + the top level instead
 *       of nchunk (the reference's loop being parallelised: cholesky.h:126-179);
 *   1 16 lanes per problem walking the chunks one after the other;
 *   0 the single-lane version (the host-checked form; on-device cross-check and A/B).
 * Widths 9..32 (and plans with general terms): 2 = a Kogge-Stone scan over composed elements when the plan has few
 * problems with many chunks (B x nchunk <= 1024 at widths <= 16, 512 above, nchunk >= 8; csrc/wide_prefix_scan.hip) --
 * such plans then choose N / 256 chunks by themselves --, else and for 1: a workgroup per problem walking the chunks. */
int clr_batch_set_prefix_mode(clr_batch* h, int mode);
/* Level structure of mode 2: `levels` (0..3) levels of groups of `group` (>= 2) elements; levels < 0 (default):
 * chosen from the chunk count by a cost model (clr_core.h: plan_prefix).  Re-plans the workspace. */
int clr_batch_set_prefix_plan(clr_batch* h, int levels, int group);
/* The plan in force: number of composition levels (0 = plain walk), group size per level [3], element count per
 * level [4] (counts[0] = chunks). */
int clr_batch_get_prefix_plan(const clr_batch* h, int* levels, int* groups, int* counts);


/* The maxima the kernel selection looks at -- max |t|, largest time step, largest |d_comp|, largest decay rate of the
 * plan's own series / coefficients -- and the host time of the last clr_batch_set_series (scan + uploads, ms).  Any
 * pointer may be NULL. */
int clr_batch_get_selection_bounds(const clr_batch* h, double* tmax, double* dxmax, double* dmax, double* cmax,
                                   double* set_series_host_ms);
/* Floors for those maxima (negative: keep): a sharded plan passes the maxima of the WHOLE batch so that every shard
 * selects the same kernels as the unsharded plan would. */
int clr_batch_set_selection_bounds(clr_batch* h, double tmax, double dxmax, double dmax, double cmax);
/* The fused log-likelihood normally needs no second pass over the series: each
 * chunk's true log-det / quadratic contributions follow from its zero-start sums and
 * its start state (determinant lemma + Woodbury), and only
 * problems with a chunk the positive-definiteness certificate or the error estimate
 * cannot settle are re-run through the exact replay.  force != 0 replays every
 * problem (the reference's recurrence step by step; for A/B measurements and as the
 * on-device cross-check).  Materialising runs always replay. */
int clr_batch_set_exact(clr_batch* h, int force);
/* After a synchronised run: how many problems went through the exact replay. */
int clr_batch_get_exact_count(clr_batch* h, int* count);
/* ... and how: level[p] = 0 settled from the chunk summaries (no second pass); 1 = the
 * conditioning record was above the bound, the chunked replay ran and every chunk's end state
 * met the scanned start state of the next chunk; 2 = settled by the truly sequential recurrence
 * (certificate failed, or the replay's end states did not meet the scanned ones).  On forced
 * exact / single-chunk runs every problem is at least 1. */
int clr_batch_get_exact_flags(clr_batch* h, int* level /* [B] */);
/* Conditioning record of the last run (widths 1..8), per problem: gamma_max = the largest
 * a_n / D_n over the zero-start pivots of its chunks (the cancellation in D_n = a_n - u.Su,
 * cholesky.h:162-175), mu_min = the smallest pivot of the chunk certificates (1 = the start
 * state does not eat into the chunk's pivots), resid_max = (after a replay) the largest
 * relative mismatch between a replayed chunk's end state and the scanned start state of the
 * next chunk.  Any pointer may be NULL. */
int clr_batch_get_conditioning(clr_batch* h, double* gamma_max, double* mu_min, double* resid_max);
/* ... the largest measured relative error eG of the chunks' G = (I + P Jm)^-1 P, per problem ... */
int clr_batch_get_measured_error(clr_batch* h, double* eg_max);
/* ... and max over chunks of gamma_c / mu_c taken chunk by chunk (diagnostic). */
int clr_batch_get_conditioning_chunkwise(clr_batch* h, double* ratio_max);
/* Routing of ill-conditioned problems.  A problem whose gamma_max / mu_min reaches
 * max_gamma_over_mu (default 1e7; <= 0: never), or whose gamma_max alone reaches the bound of
 * clr_batch_set_certificate_gamma (default 1e4), is not settled from the chunk summaries: the
 * chunked replay (the reference recurrence from the scanned start states, parallel over chunks)
 * runs for it and its end states are compared with the scanned start states; a mismatch above
 * max_residual (default 1e-11, relative) sends it to the truly sequential recurrence (one lane
 * walks the whole series; slow, exact).  Calibration: profiles/r02n_adv_probe.txt, r03_conditioning_calibration.txt. */
int clr_batch_set_certificate(clr_batch* h, double max_gamma_over_mu, double max_residual);
/* The bound on gamma_max alone (default 1e4; <= 0: no such test) -- gamma = a_n / D_n is the cancellation in the
 * recurrence itself: it is what the deviation from the reference follows (profiles/r03_conditioning_calibration.txt)
 * -- and on gamma_max x eG_max (default 3e-9; <= 0: no test), eG = the MEASURED relative accuracy of a chunk's
 * G = (I + P Jm)^-1 P (first-order forward error from the residual, computed by the correct kernels). */
int clr_batch_set_certificate_gamma(clr_batch* h, double max_gamma, double max_gamma_times_error);
/* Number of chunks the N axis is cut into for the scan (0 = auto). */
int clr_batch_set_chunks(clr_batch* h, int nchunk);
int clr_batch_get_chunks(const clr_batch* h, int* nchunk, int* chunk_len);

/* Problems whose conditioning record sends them to the checked chunked replay (route 1, see
 * clr_batch_get_exact_flags) used to be replayed inline: every chunk of such a problem sequentially, beside an otherwise
 * idle chip -- ONE borderline problem cost the whole batch a chunk-time (BASELINE configs[4]: +11 ms on 12.7).  Now the
 * evaluation leaves them pending and, before results are handed out, re-plans them as a small plan of their own with
 * many short chunks (n problems -> ~1024 / n chunks each at width 32): summarize, parallel prefix, replay of every chunk
 * from its scanned start state with the end-state check -- the same route with the same certificates (a mismatch goes to
 * the sequential recurrence, cholesky.h:176 semantics unchanged), a tenth of the chunk length.
 *   mode -1 (default): plans whose chunks hold >= 1024 samples; 0: never (the inline replay; also restores bit-identical
 *   results under any sharding for route-1 problems, whose side plan's chunking depends on how many there are per shard);
 *   1: whenever the plan has more than one chunk.  More pending problems than a quarter of the batch (or 256) are
 *   replayed inline after all.  Forced-exact and materialising runs replay everything and never defer. */
int clr_batch_set_rescue(clr_batch* h, int mode);
/* Of the last evaluation whose results were fetched: how many problems were re-planned (negative: that many were
 * replayed inline instead), the running total, and the side plan's chunking.  Any pointer may be NULL. */
int clr_batch_get_rescue(const clr_batch* h, int* last_count, long* total, int* nchunk, int* chunk_len);

/* What a materialising run (clr_batch_enqueue(h, 1), widths 1..8) keeps in HBM:
 *   0 (default) the reference's four arrays phi, u, W, D (cholesky.h:76-78; 8 N (3 J + 1) bytes per problem);
 *   1 LEAN: W and D only (8 N (J + 1) bytes per problem, SURVEY.md 8d row A-lean).  phi[:, n] = exp(-c (t_{n+1} - t_n))
 *     and u[:, n - 1] = U~(t_n) are pure functions of the times and the coefficients (cholesky.h:127-147): the replay does
 *     not store them, and clr_batch_get_factor regenerates them on the device with the very functions the replay
 *     evaluates -- W, D and u come back bit-identical to layout 0's, phi to one ulp (the exp tier of a step is chosen per
 *     wave, and the expanding kernel groups samples into waves differently from the replay).  At B = 1024, N = 1e5, width 8 the factor shrinks
 *     from 20.5 to 7.4 GB and the materialising replay is no longer bound by its stores.  The lean factor can be
 *     expanded as long as the plan still holds the series and coefficients of the materialising run (else
 *     clr_batch_get_factor returns CLR_NOT_COMPUTED). */
int clr_batch_set_factor_layout(clr_batch* h, int layout);
/* Accuracy of the materialised factor at the chunk heads (widths 1..8).  The chunked replay starts every chunk from its
 * SCANNED start state; the scan algebra's rounding in that state -- amplified by the cancellation in
 * D_n = a - u^T S u (cholesky.h:162-175) -- shows in W and D for the ~32 samples the recurrence needs to forget it:
 * 4e-11 (of the largest entry) at a chunk's first sample against 4e-13 from sample 32 on, which is the distance of the
 * reference's own sequential recurrence from a binary128 evaluation (profiles/r06d_factor_error.txt; N = 1e5, width 8).
 * With `samples` > 0 (default 64) a materialising run recomputes the first `samples` entries of every chunk from the
 * state the PREVIOUS chunk's replay reached at the boundary -- the reference recurrence carried across it -- in a
 * second, short launch (samples / chunk_len of a replay: 4 % at the bench shape).  0: round 5's factor. */
int clr_batch_set_factor_refine(clr_batch* h, int samples);
/* Bytes of factor one problem occupies in HBM under the layout and chunking in force. */
int clr_batch_get_factor_bytes(const clr_batch* h, size_t* bytes_per_problem);

/* Materialising runs (widths 1..8) as a PIPELINE over `groups` contiguous groups of problems: the summarize pass of
 * group g + 1 (fp64-VALU-bound) runs while group g is replayed (HBM-bound: the factor's stores), on streams that own
 * disjoint sets of compute units -- `summarize_cus` of the device's CUs (a multiple of 16; 0: no CU masks, the
 * dispatcher decides) for `summarize_streams` summarize streams, the others for the replay stream; the prefix and the
 * corrections of a group run on a third stream in between.  Results and the factor are those of the plain sequence
 * of kernels, bit for bit (the same kernels on the same data, group by group).  groups = 0: off (default). */
int clr_batch_set_materialize_pipeline(clr_batch* h, int groups, int summarize_cus, int summarize_streams);


/* Enqueue one evaluation of all B problems on the handle's stream (inputs
 * already resident in HBM).  `materialize` != 0 additionally writes the factor
 * (phi, u, W, D per problem; 8 N (3J+1) bytes each) to HBM, as B separate
 * CholeskySolver.compute calls would.  On the device it is kept chunk-interleaved
 * ([i][j][chunk]: every store instruction of a wave writes 512 contiguous bytes);
 * clr_batch_get_factor returns it in the reference's storage order. */
int clr_batch_enqueue(clr_batch* h, int materialize);
/* Wait for the stream. */
int clr_batch_synchronize(clr_batch* h);
/* HBM -> host; any pointer may be NULL. */
int clr_batch_get_results(clr_batch* h, double* loglike, double* logdet,
                          double* quad, int* status);
/* One optimiser / MCMC evaluation (celerite.py:160-219 per problem: new parameters -> compute -> log_likelihood) in
 * one call: clr_batch_set_coefficients + clr_batch_enqueue(h, 0) + clr_batch_get_results.  For launch-latency-sized
 * batches (BASELINE configs[1]: a 65-us kernel) the three separate calls of a Python caller cost as much as the kernel. */
int clr_batch_evaluate(clr_batch* h, const double* jitter, const double* a_real, const double* c_real,
                       const double* a_comp, const double* b_comp, const double* c_comp, const double* d_comp,
                       double* loglike, double* logdet, double* quad, int* status);
/* CholeskySolver::solve (cholesky.h:218-318) for every problem of the plan at once: x = K_p^-1 b_p from the factor of
 * the last materialising run (clr_batch_enqueue(h, 1); either factor layout), parallel in n -- forward substitution,
 * division by D and backward substitution as two chunked affine scans whose chunk maps are formed once and shared by
 * all right-hand sides and both sweeps (csrc/clr_bsolve_kernels.h).  b and x: host arrays [B][nrhs][N] (row-major:
 * right-hand side k of problem p at (p * nrhs + k) * N); b == NULL with nrhs == 1: the plan's own y (already in HBM --
 * GP.apply_inverse(y) of celerite.py:307-328 for B problems without an upload).  Widths 1..8: chunked plans (N >= 128).
 * Widths 9..64 (round 6): the factor lies in the reference's storage and the sweeps are the object API's wave-per-chunk
 * affine scans (csrc/wsweep_kernels.hip) launched once for the whole batch, grid.z = problem; N >= 512.  A problem
 * whose materialising run reported CLR_NOT_POSITIVE_DEFINITE has no factor: its x is undefined (non-finite).  With the
 * LEAN layout the sweeps regenerate phi and u from (t, coefficients) and move 9 instead of 25 doubles per sample through
 * HBM four times: the plan must still hold the series and coefficients of the materialising run. */
int clr_batch_solve(clr_batch* h, int nrhs, const double* b, double* x);
/* CholeskySolver::predict (cholesky.h:599-698: the conditional mean K_p(x*, t_p) K_p^-1 y_p behind GP.predict,
 * celerite.py:330-420) for every problem of the plan, M prediction points each: xs host [B][M] (xs_stride = M) or one
 * set of M points shared by all problems (xs_stride = 0); pred host [B][M].  alpha = K^-1 y comes from the batched solve
 * on the factor of the last materialising run (any layout, widths 1..64) and stays on the device; the two passes over
 * the prediction points are the object API's chunked diagonal scans on the plan's resident times and coefficients
 * (sorted points: parallel in n; unsorted: the sequential walk of the reference). */
int clr_batch_predict(clr_batch* h, int M, const double* xs, long xs_stride, double* pred);
/* CholeskySolver::dot_L (cholesky.h:409-431: y = L z with K = L L^T, what GP.sample draws, celerite.py:422-451) for
 * every problem of the plan from the factor of its last materialising run (either layout): z, y host [B][nrhs][N].
 * Widths 1..8: a chunked diagonal scan on the chunk-interleaved factor, lane = (problem, chunk)
 * (csrc/clr_bdotl_kernels.h); widths 9..64: the object API's wave-per-chunk scan launched once for the whole batch
 * (grid.z = problem; series shorter than 2048 samples: the sequential kernel per problem).  clr_batch_get_solve_ms
 * then reports this call's device time. */
int clr_batch_dot_L(clr_batch* h, int nrhs, const double* z, double* y);
/* CholeskySolver::dot (cholesky.h:441-596: y = K z, GP.dot of celerite.py:453-489) for every problem of the plan, K_p given
 * by the plan's resident times and the coefficients in force -- its diagonal is sum a_real + sum a_comp + jitter, the
 * observational variance is not part of it (:483-485); no factor and no materialising run are needed.  z, y host
 * [B][nrhs][N].  Widths 1..8: both triangles as chunked diagonal scans with the features evaluated on the fly, lane =
 * (problem, chunk) (csrc/clr_bdot_kernels.h); widths 9..64: the object API's kernels problem by problem. */
int clr_batch_dot(clr_batch* h, int nrhs, const double* z, double* y);
/* Device time of the last clr_batch_solve (HIP events around its kernels: relayout, the five phases, relayout back;
 * the host <-> HBM copies of b and x are outside). */
int clr_batch_get_solve_ms(const clr_batch* h, double* device_ms);
/* After a materialising run: copy problem p's factor to host buffers. */
int clr_batch_get_factor(clr_batch* h, int p, double* phi, double* u, double* W, double* D);

/* Runs `steps` evaluations back to back, bracketing every kernel with HIP
 * events recorded on the handle's stream.  kernel_ms[6] receives the SUMMED
 * device time of the relayout / summarise / prefix / correct / replay / finalise
 * kernels, total_ms the first-event-to-last-event time.  In layout 1 only:
 * with relayout_each_step != 0 the row-major -> interleaved transposition is
 * redone inside every step (the cost when every evaluation brings NEW series);
 * otherwise it is done once, outside the timed region (fixed series, new
 * hyper-parameters: the MCMC / optimiser loop of celerite.py:160-219). */
int clr_batch_run_timed(clr_batch* h, int materialize, int steps, int relayout_each_step,
                        double* total_ms, double* kernel_ms /* [6] */);

/* Per-kernel device times of the REAL loop: with profiling on, every clr_batch_enqueue
 * brackets its kernels with HIP events on the plan's stream (up to 4096 evaluations since
 * the switch); clr_batch_get_profile synchronises and returns the summed milliseconds per
 * kernel (same order as clr_batch_run_timed) and the number of evaluations recorded. */
/* on = 2: only the summarize kernel (the dominant one; the warm-started recurrence when that runs) is bracketed --
 * two event records per evaluation instead of seven (an event record costs ~5 us of stream time); widths 1..8. */
int clr_batch_set_profiling(clr_batch* h, int on);
int clr_batch_get_profile(clr_batch* h, double* kernel_ms /* [6] */, int* steps);



/* Convenience: create + set + enqueue + get + destroy, host pointers in/out. */
int clr_batch_log_likelihood(int B, int N, int J_real, int J_comp,
                             const double* jitter,
                             const double* a_real, const double* c_real,
                             const double* a_comp, const double* b_comp,
                             const double* c_comp, const double* d_comp,
                             const double* t, long t_stride,
                             const double* diag, long diag_stride,
                             const double* y, long y_stride,
                             double* loglike, double* logdet, double* quad,
                             int* status, int device);

/* Batched grad_log_likelihood (celerite/solver.cpp:347-463 per problem; no general terms): one wave per
 * (problem, partial derivative), the forward-mode tangent recurrence of csrc/grad_kernels.hip.  Host
 * pointers in and out, one shot.  value[b] = -(y^T K^-1 y + log det K + pi log N) / 2 (the reference's
 * constant, solver.cpp:415), grad is [B][1 + 2 J_real + 4 J_comp] in the order jitter, a_real, c_real,
 * a_comp, b_comp, c_comp, d_comp (d/d jitter = 0 where jitter <= DBL_EPSILON, solver.cpp:379-389);
 * status[b] = CLR_NOT_POSITIVE_DEFINITE gives value -inf and a zero gradient (GP's quiet semantics,
 * celerite.py:285-291).  Widths 1..64. */
int clr_batch_grad_log_likelihood(int B, int N, int J_real, int J_comp, const double* jitter,
                                  const double* a_real, const double* c_real, const double* a_comp,
                                  const double* b_comp, const double* c_comp, const double* d_comp,
                                  const double* t, long t_stride, const double* diag, long diag_stride,
                                  const double* y, long y_stride, double* value, double* grad, int* status,
                                  int device);

/* The same on a plan (widths 1..8, no general terms): value and gradient of every problem at the coefficients in
 * force, PARALLEL IN n (csrc/clr_grad_core.h).  The tangent recurrences of solver.cpp:347-463 are linear in the
 * tangent state once the base trajectory is fixed, so after an evaluation by the scan every (chunk, direction) runs
 * its tangent from a zero tangent state at the chunk's scanned start state; three riders of the base trajectory per
 * chunk carry the tangent states across the chunk boundaries in a walk over the chunks per direction.  Problems the
 * scan routed to the sequential recurrence take the sequential kernel above (count: clr_batch_get_grad_fallbacks).
 * Synchronous; conventions of value / grad / status as above.  clr_batch_grad_log_likelihood uses this path for
 * widths 1..8 and N >= 512.  Chunked plans of widths 9..32 (and with general terms up to a total width of 32) run the
 * same decomposition with a wave per (chunk, direction) (csrc/wide_grad_kernels.hip); round 6: chunked plans of widths
 * 33..64 too (the riders at the padded width 64 under a workgroup of 256 threads); plans of widths 33..64 with ONE chunk
 * run the sequential tangent kernel on their resident arrays (every problem counted in clr_batch_get_grad_fallbacks);
 * above 64: CLR_UNSUPPORTED (CholeskySolver.grad_log_likelihood takes any width). */
int clr_batch_grad(clr_batch* h, double* value, double* grad, int* status);
int clr_batch_get_grad_fallbacks(const clr_batch* h, int* count);
/* How clr_batch_grad differentiates.  mode 0 (default): REVERSE mode -- the riders pass also records w, D, x per
 * sample and the state every K steps, the adjoint at every chunk end follows from the riders in a walk backwards over
 * the chunks, and ONE reverse sweep per chunk yields all partials (cost ~4x an evaluation instead of ~20x; needs
 * 8 (J + 2) bytes per sample of HBM plus the stored states, falls back to mode 1 when that does not fit).  The sweep
 * reconstructs the states between the stored ones; the drift it measures at every stored state certifies them, and a
 * problem that drifts beyond `drift_tolerance` (default 1e-9; <= 0 keeps the current one) is redone in mode 1.
 * mode 1: FORWARD mode, one tangent per partial (the description above).  stored_state_distance: a state every K steps;
 * 0 (default) = wherever the decay accumulated since the last stored state reaches the growth budget (c T <= 3 over any
 * rebuilt stretch: a few states per chunk on dense series, one every few samples on sparse ones).
 * In reverse mode the three riders of a chunk come from the scan's own element of that chunk (one J x J solve per chunk
 * instead of ~190 FMAs per sample) whenever a gradient chunk is a scan chunk; the sweep's second certificate -- the
 * adjoint it arrives at for the chunk's first sample against the one predicted from the riders -- vouches for them.
 * mode 2 = reverse mode with the riders accumulated along the trajectory (A/B runs). */
int clr_batch_set_grad_mode(clr_batch* h, int mode, int stored_state_distance, double drift_tolerance);
/* Of the last clr_batch_grad: whether the reverse sweep ran, how many problems were redone in forward mode, the
 * largest drift among the problems the reverse sweep settled. */
int clr_batch_get_grad_info(const clr_batch* h, int* reverse_used, int* forward_reruns, double* drift_max);

/* ---- the batch axis over several GPUs (BASELINE config 4) ---------
 * Problems are independent -- every member of the reference solver is per object
 * (cholesky.h:703-706) -- so the batch axis shards embarrassingly: shard s of S owns the
 * contiguous slice clr_shard_bounds(B, S, s) and is an ordinary clr_batch plan on
 * devices[s], driven by its own host thread (stream + pinned staging per shard).  No
 * collective, no device-to-device traffic; results are concatenated on the host.
 * A device may be listed more than once (the shards then share it).
 * Results do not depend on the sharding, BIT FOR BIT, with the default settings, as long as every
 * sharding uses the same chunk count (clr_sharded_set_chunks; the automatic choice looks at a shard's
 * batch size -- what fills one GPU): every decision a plan takes from a count over its problems --
 * kernel selection, the prefix plan, the warm-started recurrence's activation and adaptation, side
 * plan or inline replay of level-1 problems and the side plan's chunk count -- is taken once, from
 * the counts over the WHOLE batch (csrc/sharded.cpp, csrc/clr_group_hooks.h). */
typedef struct clr_sharded clr_sharded;

/* [lo, hi) of `shard` when `total` problems are cut into `nshards` contiguous slices
 * (the first total % nshards slices are one longer).  Pure host arithmetic. */
int clr_shard_bounds(int total, int nshards, int shard, int* lo, int* hi);
/* nshards = number of entries of `devices` (clamped to B).  NULL on failure:
 * clr_sharded_last_error() says why (no device, bad index, allocation). */
clr_sharded* clr_sharded_create(int B, int N, int J_real, int J_comp, const int* devices, int nshards);
void clr_sharded_destroy(clr_sharded* h);
const char* clr_sharded_last_error(void);
int clr_sharded_num_shards(const clr_sharded* h);
int clr_sharded_get_shard(const clr_sharded* h, int shard, int* device, int* lo, int* hi);
/* As the clr_batch_* entries of the same name, over the whole batch (host arrays of B
 * problems; every shard takes its slice).  The calls return when every shard has. */
int clr_sharded_set_chunks(clr_sharded* h, int nchunk);
int clr_sharded_get_chunks(const clr_sharded* h, int shard, int* nchunk, int* chunk_len);
/* (the automatic choice of the summarize kernel is resolved once for the whole batch: the shards are handed the
 * batch-wide maxima of the series and coefficients, so the choice does not depend on the sharding) */
int clr_sharded_set_summarize_mode(clr_sharded* h, int mode);
/* clr_batch_set_warm_start on every shard.  The warm-started recurrence is switched on when at least half of the
 * problems OF THE BATCH are eligible and lengthens its warm-ups by the fallbacks OF THE BATCH (the shards' counts are
 * added up between the two halves of every evaluation's resolve): the same route under any sharding.  Its chunking
 * follows the chunk count when that is pinned, else the shard's batch size. */
int clr_sharded_set_warm_start(clr_sharded* h, int mode, int forced_warmup);
/* clr_batch_set_rescue on every shard.  Side plan or inline replay, and the side plan's chunk count, follow the number
 * of route-1 problems of the WHOLE batch: bit-identical under any sharding (round 5: per shard).
 * clr_sharded_get_rescue: problems of the last fetched evaluation that took the checked route outside the main pass,
 * summed over the shards. */
int clr_sharded_set_rescue(clr_sharded* h, int mode);
int clr_sharded_get_rescue(const clr_sharded* h, int* last_count);
/* clr_batch_set_certificate + clr_batch_set_certificate_gamma on every shard (the routing bounds of ill-conditioned
 * problems; a negative max_gamma / max_gamma_times_error leaves the gamma bounds as they are). */
int clr_sharded_set_certificate(clr_sharded* h, double max_gamma_over_mu, double max_residual, double max_gamma,
                                double max_gamma_times_error);
/* The summarize kernel all shards will run (clr_batch_get_summarize_kernel; -1 if they disagree). */
int clr_sharded_get_summarize_kernel(const clr_sharded* h, int* kind);
/* clr_batch_get_series_order / clr_batch_clear_series over all shards (the smallest step of the whole batch) */
int clr_sharded_get_series_order(const clr_sharded* h, double* dtmin);
int clr_sharded_clear_series(clr_sharded* h);
int clr_sharded_set_series(clr_sharded* h, const double* t, long t_stride, const double* diag,
                           long diag_stride, const double* y, long y_stride);
int clr_sharded_set_coefficients(clr_sharded* h, const double* jitter, const double* a_real,
                                 const double* c_real, const double* a_comp, const double* b_comp,
                                 const double* c_comp, const double* d_comp);
int clr_sharded_enqueue(clr_sharded* h);
int clr_sharded_synchronize(clr_sharded* h);
int clr_sharded_get_results(clr_sharded* h, double* loglike, double* logdet, double* quad, int* status);
/* clr_batch_grad on every shard concurrently (coefficients in force: clr_sharded_set_coefficients); grad is
 * [B][1 + 2 J_real + 4 J_comp]. */
int clr_sharded_grad(clr_sharded* h, double* value, double* grad, int* status);
/* One optimiser / MCMC evaluation: new coefficients in, B log-likelihoods out
 * (set_coefficients + enqueue + get_results on every shard concurrently). */
int clr_sharded_evaluate(clr_sharded* h, const double* jitter, const double* a_real, const double* c_real,
                         const double* a_comp, const double* b_comp, const double* c_comp,
                         const double* d_comp, double* loglike, double* logdet, double* quad, int* status);
/* The consumers of the factor on a sharded batch (GP.apply_inverse / .sample / .predict for B problems over several
 * GPUs): clr_sharded_materialize runs clr_batch_enqueue(plan, 1) on every shard, settles the evaluation with the
 * batch-wide counts (results as clr_sharded_get_results; any pointer may be NULL) and leaves every shard's factor in
 * its HBM; clr_sharded_solve / _dot_L / _dot / _predict are clr_batch_solve / _dot_L / _dot / _predict on every shard concurrently,
 * each on its contiguous slice of the host arrays ([B][nrhs][N]; xs [B][M] or shared with xs_stride = 0).  No
 * collective: every problem's state is its own (cholesky.h:703-706). */
int clr_sharded_materialize(clr_sharded* h, double* loglike, double* logdet, double* quad, int* status);
int clr_sharded_solve(clr_sharded* h, int nrhs, const double* b, double* x);
int clr_sharded_dot_L(clr_sharded* h, int nrhs, const double* z, double* y);
int clr_sharded_dot(clr_sharded* h, int nrhs, const double* z, double* y);
int clr_sharded_predict(clr_sharded* h, int M, const double* xs, long xs_stride, double* pred);
/* `steps` back-to-back evaluations on every shard concurrently (HIP events per shard);
 * shard_ms[s] = that shard's first-to-last event time. */
int clr_sharded_run_timed(clr_sharded* h, int steps, double* shard_ms);
/* clr_batch_log_likelihood over `ndevices` shards (create + set + evaluate + destroy). */
int clr_batch_log_likelihood_sharded(int B, int N, int J_real, int J_comp, const double* jitter,
                                     const double* a_real, const double* c_real, const double* a_comp,
                                     const double* b_comp, const double* c_comp, const double* d_comp,
                                     const double* t, long t_stride, const double* diag, long diag_stride,
                                     const double* y, long y_stride, double* loglike, double* logdet,
                                     double* quad, int* status, const int* devices, int ndevices);

/* ---- celerite::carma::CARMASolver (cpp/include/celerite/carma.h; solver.cpp:200-235) ----
 * The reference's comparison solver: a CARMA(p, q) process in carma_pack's parameterisation, its log-likelihood by
 * a Kalman filter, and the conversion to celerite coefficients (tests/test_celerite.py:22-42 checks that the two
 * likelihoods agree).  The model algebra runs on the host at creation; the filter -- sequential in n -- runs as one
 * wave on the device.  Errors: q >= p -> CLR_DIMENSION_MISMATCH (carma.h:59); p > CLR_CARMA_MAX_ORDER ->
 * CLR_UNSUPPORTED; a negative predicted variance -> CLR_CARMA_INSTABILITY (carma.h:185-186). */
typedef struct clr_carma clr_carma;
/* CARMASolver(log_sigma, arparams[p], maparams[q]), carma.h:54-72.  NULL on error (see clr_last_error). */
clr_carma* clr_carma_create(double log_sigma, int p, const double* arparams, int q, const double* maparams,
                            int* status);
void clr_carma_destroy(clr_carma* h);
/* CARMASolver::log_likelihood(t, y, yerr), carma.h:221-239; host pointers, copied in. */
int clr_carma_log_likelihood(clr_carma* h, int n_t, const double* t, int n_y, const double* y, int n_yerr,
                             const double* yerr, double* out);
/* CARMASolver::get_celerite_coeffs, carma.h:74-139.  Counts first (pass NULL arrays), then the values: a_real,
 * c_real [n_real]; a_comp, b_comp, c_comp, d_comp [n_comp].  No device work. */
int clr_carma_get_celerite_coeffs(const clr_carma* h, int* n_real, int* n_comp, double* a_real, double* c_real,
                                  double* a_comp, double* b_comp, double* c_comp, double* d_comp);

/* ---- O(J) host helpers the Python layer imports (celerite/terms.py:18) --------
 * Plain host C++ (no device work): cpp/include/celerite/utils.h:106-163,27-104. */
double clr_kernel_value(int J_real, const double* a_real, const double* c_real,
                        int J_comp, const double* a_comp, const double* b_comp,
                        const double* c_comp, const double* d_comp, double tau);
double clr_psd_value(int J_real, const double* a_real, const double* c_real,
                     int J_comp, const double* a_comp, const double* b_comp,
                     const double* c_comp, const double* d_comp, double omega);
int clr_check_coefficients(int n_a_real, const double* a_real, int n_c_real, const double* c_real,
                           int n_a_comp, const double* a_comp, int n_b_comp, const double* b_comp,
                           int n_c_comp, const double* c_comp, int n_d_comp, const double* d_comp);

#ifdef __cplusplus
}
#endif
#endif /* CELERITE_HIP_H */
