// celerite_amd/csrc/clr_generic_kernels.h
//
// Any-width (J <= CLR_MAX_WIDTH, including "general" A/U/V rows) kernels for the
// single-problem solver object.  These cover what the fixed-width scan kernels
// do not: widths above 8, general semiseparable terms, and every consumer of the
// stored factor (dot_solve / solve / dot_L / dot / predict).  They are
// sequential in n like the reference and parallel across the width:
//   factor_generic   one 256-thread workgroup per problem; S (J x J) in LDS.
//   the sweep kernels   one 64-lane wave per right-hand side; lane j (and j+64)
//                       carries row j of the running J-vector f, the inner
//                       products are wave reductions (DPP/shuffle).
// Loads of the factor are coalesced along j (reference storage [j + J n]).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace clr {

struct GenericProblem {
  int N, J, J_real, J_comp, J_general;
  const double *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp;  // device
  const double *U, *V;  // device, row-major [J_general][N] (may be null)
  const double* t;      // device [N]
};

// A batch of problems WITH general terms (cholesky.h:65-72,148-152; clr_batch_set_general): one workgroup per
// problem, sequential in n exactly as the reference, compute (cholesky.h:100-179) fused with dot_solve
// (:343-357) and log_determinant; nothing is stored but the four results per problem.
struct GenericBatch {
  int B, N, J_real, J_comp, J_general;
  const double *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp, *jitter;  // device, [B][...] / [B]
  const double *t, *diag, *y;
  long t_stride, diag_stride, y_stride;  // per problem (0: one shared series)
  const double *A, *U, *V;               // A [N], U / V row-major [J_general][N] per problem
  long A_stride, U_stride, V_stride;     // per problem (0: shared)
  double *out_ll, *out_logdet, *out_quad;
  int* out_status;
};
void launch_generic_loglike_batch(const GenericBatch& G, hipStream_t s);

// dot_solve / solve on a stored factor as chunked scans over n (sweep_kernels.hip).
struct SweepParams {
  int N, J, nchunk, L, nrhs;
  const double *phi, *u, *W, *D;  // the factor, reference storage
  const double* in;               // forward: b; backward: the forward sweep's output (undivided)
  double* out;                    // x per column (may alias `in` going backward) or null
  double* quad;                   // [nrhs] sum x_n^2 / D_n (dot_solve) or null
  int backward;
  double *elems, *starts, *part;  // workspace, set by launch_sweep_scan
  // two-level prefix of the wave-per-chunk sweeps (wsweep_kernels.hip): runs of run_len chunks, the state at each run's start
  int run_len;                    // 0: one walk over all chunks
  const double* run_starts;       // [nrhs][runs][J + 1] or null (the series' first sample)
  // a BATCH of problems in one launch (clr_batch_solve on the wide plans' factors: wsweep_kernels.hip, grid.z = problem):
  // problem b's arrays start b strides further on; 0 / 1 problems: the single solver
  int batch;
  long stride_phi, stride_W, stride_D;  // doubles between two problems' phi / u, W, D
  long stride_in, stride_out;           // ... right-hand sides and results (nrhs * N each)
  long stride_ws;                       // ... workspaces (wsweep_workspace_doubles)
};
bool sweep_scan_supported(int N, int J);
int sweep_chunks(int N);
size_t sweep_workspace_doubles(int J, int nchunk, int nrhs);
void launch_sweep_scan(SweepParams P, double* workspace, hipStream_t s);
// N >= 2048, width <= 32 (wsweep_kernels.hip): wave per chunk, lane = column of the chunk's affine map
bool wsweep_scan_supported(int N, int J);
int wsweep_chunks(int N, int J);
size_t wsweep_workspace_doubles(int J, int nchunk, int nrhs);
void launch_wsweep_scan(SweepParams P, double* workspace, hipStream_t s);
// dot_L at N >= 2048, any width: wave per chunk, lane = row; workspace nrhs * nchunk * 3 J doubles
bool wdotl_scan_supported(int N, int J);
int wdotl_chunks(int N);
void launch_wdotl_scan(SweepParams P, double* workspace, hipStream_t s);
// dot (cholesky.h:533-560) the same way; P.phi / P.u = dot's own phi, u (J x N), v J x N, dg the diagonal
void launch_wdot_scan(SweepParams P, const double* v, const double* dg, double* workspace, hipStream_t s);
// dot_L (cholesky.h:409-431) as a chunked diagonal scan; workspace: nrhs * nchunk * 3 J doubles
void launch_dot_L_scan(SweepParams P, double* workspace, hipStream_t s);
// predict (cholesky.h:599-698): chunked diagonal scans + one thread per (sorted) prediction point
bool predict_scan_supported(int N, int J_real, int J_comp);
size_t predict_workspace_doubles(int nchunk, int rows);
void launch_predict_scan(const GenericProblem& g, const double* alpha, int M, const double* xs, double* pred,
                         double* workspace, int nchunk, int L, hipStream_t s);

// solver.cpp:347-463 (grad_kernels.hip): one wave per partial derivative.
struct GradParams {
  int N, J_real, J_comp, J_general;
  const double *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp;  // device
  double jitter;
  const double *A, *U, *V;      // device; A null without general terms; U, V row-major [J_general][N]
  const double *t, *diag, *y;   // device [N]
  int fast_trig;                // host-verified: max|d_comp| * max|t| < CLR_FAST_TRIG_LIMIT
  double* out_value;            // [1]  -(quad + log det + pi log N) / 2
  double* out_grad;             // [1 + 2 J_real + 4 J_comp]
  int* out_status;              // [1]
  // batched form (clr_batch_grad_log_likelihood; no general terms): blockIdx.y = problem b reads its own
  // coefficient rows ([B][J_real] / [B][J_comp]), jitter_b[b], series b * stride, and writes value[b],
  // grad[b][...], status[b].  B == 0: the single-problem call above.
  int B;
  const double* jitter_b;
  long t_stride, diag_stride, y_stride;
  const int* only_level;  // batched, may be null: only the problems with only_level[b] >= 2 are computed
  // CHUNKED mode (the gradient parallel in n at widths 9..32 and with general terms, wide_grad_kernels.hip):
  // blockIdx.z = chunk c.  The wave runs samples [chunk c) only, from the TRUE base state at the chunk's first sample
  // (starts: [B][nchunk][JP (JP + 1) / 2 + JP] from the wide scan, packed upper triangle at the padded width JP | f)
  // and a ZERO tangent state, and writes the record of (problem b, chunk c, direction p) to
  // rec + ((b nchunk + c) NG + p) (JP JP + JP + 2):  dS_end [JP][JP] | df_end [JP] | d log det | d quad.
  // only_level (if given): only the problems with only_level[b] < 2 (the scan's start states are certified for them).
  int nchunk, L, L0, JP;
  const double* starts;
  double* rec;
  long A_stride, U_stride, V_stride;  // general terms per problem (0: shared)
};
void launch_grad(const GradParams& P, hipStream_t s);
void launch_grad_chunked(const GradParams& P, hipStream_t s);  // (the batched form, P.nchunk >= 2)
// one problem (P.B == 0) at any total width up to CLR_MAX_WIDTH_ANY (grad_any_kernels.hip): one workgroup per direction,
// S and dS in `workspace` (grad_any_workspace_doubles); non-zero: the kernel could not be configured
size_t grad_any_workspace_doubles(int J, int NG);
int launch_grad_any(const GradParams& P, double* workspace, hipStream_t s);

// The other two kernels of the chunk-parallel gradient at the padded widths JP = 16 / 32 (wide_grad_kernels.hip):
struct WideGradWalk {
  int B, NG, nchunk, JP, N;
  const double* elems;    // [B][nchunk][ELEM(JP)]  the wide scan's chunk elements (A | b | C | eta | Jm)
  const double* starts;   // [B][nchunk][START(JP)]
  double* riders;         // [B][nchunk][JP JP (AA) + JP (eta) + JP JP (JJ)]
  const double* rec;      // [B][nchunk][NG][JP JP + JP + 2]
  const int* level;       // [B] route of the evaluation (may be null); problems with level >= 2 are skipped
  const double* ll;       // [B] log-likelihood of the evaluation
  const int* ll_status;   // [B]
  const double* jitter;   // [B]
  double* out_value;      // [B]  -(quad + log det + pi log N) / 2   (solver.cpp:415)
  double* out_grad;       // [B][NG]
  int* out_status;        // [B]
};
int launch_wide_grad_riders(const WideGradWalk& W, hipStream_t s);  // riders of every (problem, chunk) from element + start; non-zero: LDS not configurable
int launch_wide_grad_walk(const WideGradWalk& W, hipStream_t s);    // one wave per (problem, direction): walk the chunks

// cholesky.h:41-210.  D must arrive initialised to the full diagonal
// (diag + sum a_real + sum a_comp + jitter [+ A], cholesky.h:98-99).
// status[0] = 1 when some D_n < 0 (n >= 1), log_det[0] = sum log D_n.
void launch_factor_generic(const GenericProblem& g, double* phi, double* u, double* W,
                           double* D, int* status, double* log_det, hipStream_t s);

// Builds phi, u, v for `dot` (cholesky.h:487-531); v is J x N.
void launch_dot_setup(const GenericProblem& g, double* phi, double* u, double* v,
                      hipStream_t s);

// cholesky.h:326-401: out[0] = b^T K^-1 b.
void launch_dot_solve(int N, int J, const double* phi, const double* u, const double* W,
                      const double* D, const double* b, double* out, hipStream_t s);
// cholesky.h:218-318, column-major (N, nrhs).
void launch_solve(int N, int J, int nrhs, const double* phi, const double* u, const double* W,
                  const double* D, const double* b, double* x, hipStream_t s);
// cholesky.h:409-431.
void launch_dot_L(int N, int J, int nrhs, const double* phi, const double* u, const double* W,
                  const double* D, const double* z, double* y, hipStream_t s);
// cholesky.h:533-560: y = K z given phi, u, v and the constant diagonal dg[N].
void launch_dot(int N, int J, int nrhs, const double* phi, const double* u, const double* v,
                const double* dg, const double* z, double* y, hipStream_t s);
// cholesky.h:599-698 given alpha = K^-1 y.
void launch_predict(const GenericProblem& g, const double* alpha, int M, const double* xs,
                    double* pred, hipStream_t s);
// widths above CLR_MAX_WIDTH up to CLR_MAX_WIDTH_ANY (huge_kernels.hip): S in HBM, one workgroup per problem / right-hand side
size_t factor_huge_workspace_doubles(int J);
void launch_factor_huge(const GenericProblem& g, double* S, double* phi, double* u, double* W, double* D, int* status,
                        double* log_det, hipStream_t s);
void launch_dot_solve_huge(int N, int J, const double* phi, const double* u, const double* W, const double* D,
                           const double* b, double* out, hipStream_t s);
void launch_solve_huge(int N, int J, int nrhs, const double* phi, const double* u, const double* W, const double* D,
                       const double* b, double* x, hipStream_t s);
// dot_solve / solve at widths 65 .. 1024 on long series (bigsweep_kernels.hip): chunked affine scans whose J x J chunk maps
// are built once per factor and direction (maps: bigsweep_maps_doubles per direction) and shared by all right-hand sides
bool bigsweep_supported(int N, int J);
int bigsweep_chunks(int N, int J);
size_t bigsweep_maps_doubles(int J, int nchunk);
size_t bigsweep_workspace_doubles(int J, int nchunk, int nrhs);
void launch_bigsweep_maps(int N, int J, int nchunk, int L, int backward, const double* phi, const double* u, const double* W, double* maps, hipStream_t s);
void launch_bigsweep_scan(const SweepParams& P, const double* maps, double* workspace, hipStream_t s);  // P.in != P.out
// widths 33 .. 1024 (rows_kernels.hip): S in the registers of 1 / 4 / 16 / 64 workgroups, one counter barrier per step
bool factor_rows_supported(int J);
size_t factor_rows_workspace_doubles(int J);
void launch_factor_rows(const GenericProblem& g, int fast_trig /* host-verified: max|d_comp| max|t| < CLR_FAST_TRIG_LIMIT */,
                        const double* y /* device [N] or null: log_det[1] = y^T K^-1 y */, double* workspace, double* phi, double* u, double* W,
                        double* D, int* status, double* log_det /* [2] */, hipStream_t s);
bool factor_rows_batch_supported(int J_total);
void launch_factor_rows_batch(const GenericBatch& G, int fast_trig, hipStream_t s);
// J == 0 and small element-wise helpers.
void launch_diag_only(int N, const double* diag, double jitter, double* D, double* log_det,
                      hipStream_t s);

}  // namespace clr
