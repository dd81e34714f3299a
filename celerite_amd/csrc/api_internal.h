// celerite_amd/csrc/api_internal.h -- shared by the translation units of the C ABI (include/celerite_hip.h):
//   api_misc.hip    library / device entries, the CARMA handle
//   api_solver.hip  clr_solver_*: the object API (one problem per handle)
//   api_batch.hip   clr_batch_*: plans, HBM residency, path selection, evaluation, results
//   api_grad.hip    clr_batch_grad*: the gradient entry points
//   api_kernels.hip the small kernels the plans launch themselves (finalize, relayouts, factor de-interleave)
// Here: error reporting, the device buffer, the two handle structs, and the functions that turn a plan's state into
// kernel parameters (used by the evaluation and by the gradient).  Everything is internal to libcelerite_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/celerite_hip.h"
#include "../../include/celerite_hip_debug.h"
#include "clr_batch_kernels.h"
#include "clr_carma.h"
#include "clr_generic_kernels.h"
#include "clr_series_io.h"
#include "clr_small.h"
#include "clr_wide.h"
#include "clr_options.h"

// the last error message of the calling thread (clr_last_error) and its current device: ONE object per thread for the
// whole library (defined in api_misc.hip)
extern thread_local std::string clr_api_last_error;
extern thread_local int clr_api_device;
#define g_last_error clr_api_last_error
#define g_device clr_api_device

namespace {


int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(CLR_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

int visible_gfx950() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess &&
        strncmp(prop.gcnArchName, "gfx950", 6) == 0)
      ++ok;
  }
  return ok;
}

int require_device(int device) {
  static int count = -1;
  if (count < 0) count = visible_gfx950();
  if (count <= 0)
    return fail(CLR_NO_DEVICE,
                "no gfx950 (MI355X) device is visible; libcelerite_hip has no CPU path");
  if (device < 0 || device >= count) return fail(CLR_INVALID_ARGUMENT, "bad device index");
  HIP_TRY(hipSetDevice(device));
  return CLR_OK;
}

// Grow-only device buffer.
struct DevBuf {
  double* p = nullptr;
  size_t cap = 0;  // doubles
  int reserve(size_t n) {
    if (n <= cap && p) return CLR_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(n, 1);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(double)));
    cap = want;
    return CLR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

int upload(DevBuf& buf, const double* host, size_t n, hipStream_t s) {
  int st = buf.reserve(n);
  if (st != CLR_OK) return st;
  if (n) HIP_TRY(hipMemcpyAsync(buf.p, host, n * sizeof(double), hipMemcpyHostToDevice, s));
  return CLR_OK;
}

// max |t| over a series (NaN sticks).  A full O(N) pass, not the two ends: the C ABI does
// not require sorted times (GP.compute(check_sorted=False) reaches it unsorted), and the
// fast sincos is only valid for |d t| < CLR_FAST_TRIG_LIMIT at EVERY sample.
double max_abs(const double* x, long n) {
  double m = 0.0;
  for (long i = 0; i < n; ++i) {
    const double a = fabs(x[i]);
    if (!(a <= m)) m = a;
  }
  return m;
}

// Chunk count for the scan on `B` problems of `N` samples of width `J`.
int auto_chunks(int B, int N, int J, bool with_replay = false) {
  if (N < 128) return 1;
  // Cost model (measured on MI355X): the big kernels run
  // ceil(B * ceil(nchunk / 64) / 1024) rounds of waves (one per SIMD) over L = N / nchunk
  // steps at ~2.4 us per step at width 8 (3.3 us when the replay pass runs too), less at
  // smaller widths; the prefix phase costs what its plan says (clr_core.h: plan_prefix --
  // a walk over the chunks, or a multi-level prefix when the chip has room for it).  Many
  // problems want exactly one wave per SIMD; a single long series wants many short chunks.
  const double w = (0.25 + 0.75 * J * J / 64.0);
  const double c_step = (with_replay ? 3.3e-6 : 2.4e-6) * w;
  const long max_by_len = std::max<long>(1, N / 16);
  auto cost = [&](long nc) {
    long L = (N + nc - 1) / nc;
    if (nc > 1 && L > 8) L = (L + 7) & ~7L;
    const long waves = (long)B * ((nc + 63) / 64);
    const long rounds = (waves + 1023) / 1024;
    return rounds * L * c_step + clr::plan_prefix((int)nc, -1, 0, B, J).time_us * 1e-6 + (nc > 1 ? 15e-6 : 0.0);
  };
  long best = 1;
  double best_cost = cost(1);
  auto consider = [&](long nc) {
    if (nc < 1 || nc > max_by_len) return;
    const double c = cost(nc);
    if (c < best_cost) { best_cost = c; best = nc; }
  };
  for (long nc : {2L, 3L, 4L, 6L, 8L, 12L, 16L, 24L, 32L, 48L}) consider(nc);
  for (long nc = 64; nc <= max_by_len && nc <= 65536; nc += 64) consider(nc);
  return (int)best;
}

}  // namespace

/* ======================================================================== */
struct clr_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  bool have_stream = false;
  // grad_log_likelihood parallel in n (widths 1..8, no general terms): a one-problem plan kept between calls, and
  // the series it holds (an optimiser calls with the same t, diag, y and new coefficients)
  struct clr_batch* grad_plan = nullptr;
  int grad_N = 0, grad_JR = -1, grad_JC = -1;
  bool grad_wide = false, grad_had_general = false;
  std::vector<double> grad_series;
  int computed = 0, N = 0, J = 0;
  double log_det = 0.0;
  int J_real = 0, J_comp = 0, J_general = 0;
  DevBuf phi, u, W, D;                  // the factor (reference layout)
  DevBuf coeffs;                        // a_real c_real a_comp b_comp c_comp d_comp
  DevBuf t, U, V;                       // inputs kept for predict / dot
  DevBuf scratch, scratch2, scalars;    // right-hand sides, results
  DevBuf dot_buf[11];                            // clr_solver_dot's own buffers (it must not disturb a computed factor)
  DevBuf pred_buf[2];                            // clr_solver_predict: the prediction points, the predictions
  DevBuf ws_elems, ws_starts, ws_part, ws_cond;  // scan workspace
  DevBuf ws_lvl_elems, ws_lvl_starts;            // upper levels of the multi-level prefix; level buffers of the wide parallel prefix
  DevBuf gradbuf;                       // grad_log_likelihood staging
  DevBuf gradws;                        // ... above width 64: S and dS of a round of directions (grad_any_kernels.hip)
  // The factor's chunk heads (clr_batch_kernels.h, BatchParams::ends): compute() writes the factor from the SCANNED
  // start states, whose rounding shows in W and D for the ~32 samples the recurrence needs to forget it -- with the ~50
  // to 100-sample chunks of one long series that is most of the factor (N = 1e5, width 8: W 1.1e-10, solve 1.8e-10 of
  // the oracle, profiles/r06n_solver_factor_error.txt).  log_determinant and the hinted dot_solve do not depend on it,
  // so GP.log_likelihood pays nothing: the FIRST call that reads the factor (solve, dot_solve of another vector, dot_L,
  // predict, __getstate__) replays every chunk once more from the state the previous chunk's replay reached.
  DevBuf keep_diag, keep_jitter, ws_ends;  // compute's diag / jitter (kept for that pass), the chunks' end states
  clr::BatchParams refine_P;
  int refine_pending = 0;               // 0 nothing to do, 1 narrow plan kernels, 2 wide kernels
  // chunk maps of the affine scans over the stored factor at widths above 64 (bigsweep_kernels.hip): per direction, built by
  // the first sweep that needs them, dropped with the factor
  DevBuf big_maps[2];
  bool big_maps_valid[2] = {false, false};
  // the route the last compute took through the chunked flow (clr_solver_debug_route): device pointers into the workspace
  const int* route_level = nullptr;
  const double* route_cond = nullptr;
  int route_nchunk = 0;
  std::vector<double> host_coeffs;      // staging of the last upload (kept alive: async copy)
  // clr_solver_hint_rhs: the right-hand side the caller is about to pass to dot_solve; the next compute
  // folds b^T K^-1 b into its own pass over the series and dot_solve returns it for that very vector
  std::vector<double> host_rhs;
  DevBuf rhs;
  bool rhs_hint = false, have_quad = false;
  bool coeffs_lazy = false;             // host_coeffs not yet on the device (the one-launch route passes them as arguments)
  double cached_quad = 0.0;
  int* ws_flags = nullptr;
  size_t ws_flags_cap = 0;
  int* d_status = nullptr;
  // pinned staging of compute's inputs and results: copies from pageable memory cost ~15 us of host time each
  // (six of them per GP.log_likelihood on a short series: profiles/r02zzz_config0_hip_trace.txt)
  double* pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
};

struct clr_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  int B = 0, N = 0, J_real = 0, J_comp = 0, J = 0;
  int nchunk = 1, L = 0;
  int L0 = 0;                      // wide plans: samples of the first chunk when it is longer than L (0: uniform)
  int general_route = -1;          // plans with general terms: -1 the wide kernels when the total width allows, 1 the any-width sequential kernel
  int small_mode = -1;             // one-launch evaluation of short narrow problems: -1 auto, 0 off, 1 whenever supported
  // general terms through the wide kernels (widths J + J_general <= 64): their own chunking and workspace
  int gen_nchunk = 0, gen_L = 0, gen_L0 = 0;
  DevBuf gen_elems, gen_starts, gen_part, gen_cond, gen_scan;
  size_t gen_scan_ws_doubles = 0, scan_ws_doubles = 0;  // workspace of the wide parallel prefix (0: sequential walk)
  int* gen_flags = nullptr;
  bool pipeline_pinned = false;    // the caller tuned the scan pipeline (chunks, prefix, summarize kernel, layout, certificate): auto small mode stays out
  double wide_first_ratio64 = 1.45;  // ... at widths 33..64 (2.3 before the summarize split the features' work between a row's lanes: profiles/r05k_wide_feature_batch.txt)
  double wide_first_ratio = 1.25;  // wide plans: cost of a chunk with riders / cost of the riderless first chunk (1.1-1.25 within 2 %: profiles/r04c, r04p)
  const clr::BatchLaunchers* launch = nullptr;
  DevBuf coeffs, t, diag, y;          // coefficients (| jitter at the end); series in the API's row-major layout
  double* pin = nullptr;              // pinned host staging: coefficient uploads, result downloads
  size_t pin_cap = 0;
  DevBuf tT, dT, yT;                  // chunk-interleaved copies the kernels read
  long t_stride = 0, diag_stride = 0, y_stride = 0;
  int layout = 2;                     // 0 row-major direct, 1 interleaved copy, 2 staged through LDS
  double tmax = 0.0, dmax = 0.0;      // max |t|, max |d_comp| (host side, O(B))
  double dxmax = 0.0, cmax = 0.0;     // max |t[n+1] - t[n]|, max decay rate: the lazy-decay kernels need cmax * dxmax < 2^-7
  // floors for the four maxima above (clr_batch_set_selection_bounds): a sharded plan hands every shard the maxima
  // of the WHOLE batch, so that all shards pick the same kernels whatever the sharding
  double floor_tmax = 0.0, floor_dxmax = 0.0, floor_dmax = 0.0, floor_cmax = 0.0;
  double set_series_host_ms = 0.0;    // host time of the last clr_batch_set_series (scan + uploads)
  double dtmin = 0.0;                 // smallest step of t over the plan's series (negative: not sorted; NaN: a NaN time)
  clr::UploadStaging staging;         // pinned staging + copy streams of clr_batch_set_series (large series only)
  DevBuf scan;                        // results of the device-side scans of t
  int force_library_trig = 0;
  int coop_prefix = 2;                // 0 single lane, 1 16 lanes walking the chunks, 2 multi-level (clr_prefix_kernels.h)
  int plan_levels = -1, plan_g = 0;   // clr_batch_set_prefix_plan: < 0 = chosen by clr::plan_prefix
  clr::PrefixPlan plan;
  DevBuf lvl_elems, lvl_starts;       // composed elements / start states of the upper levels
  DevBuf g_riders, g_out, g_res;      // chunk-parallel gradient (clr_grad_kernels.h): riders, records, result (+ fallback)
  int grad_rebuild_span = 4;          // reverse mode, adaptive rule: stored states at least this many steps apart (GradStore::span)
  DevBuf g_rec, g_ck;                 // reverse mode: w, D, x per sample; stored states (GradStore, clr_grad_core.h)
  unsigned char* g_ckflag = nullptr;  // what the forward pass did before each step, per wave of 64 chunks
  size_t g_ckflag_cap = 0;
  std::vector<double> host_cmax;      // per problem: largest decay rate (sizes the stored states)
  std::vector<double> grad_span;      // per problem (one entry when the series is shared): longest time a scan chunk spans
  bool grad_span_valid = false;
  int grad_mode = 0;                  // clr_batch_set_grad_mode: 0 auto (reverse), 1 forward (one tangent per partial)
  int grad_K = 0;                     // > 0: distance of the stored states (steps), else from c_max dt_max
  int grad_riders_mode = 0;           // 0 auto (from the scan's elements when a gradient chunk is a scan chunk), 1 along the trajectory
  double grad_drift_tol = 1e-9;       // a reverse sweep whose reconstructed states drift further is redone forward
  double grad_drift_max = 0.0;        // last gradient: largest drift among the problems it settled
  int grad_forward_reruns = 0;        // ... and the problems redone by the forward-mode kernels
  bool grad_reverse_used = false;
  std::vector<double> host_jitter;    // per problem, as set (the reference zeroes d/d jitter at jitter <= eps)
  bool grad_scan_only = false;        // the evaluation inside clr_batch_grad: by the scan, its start states are needed
  int grad_fallbacks = 0;             // problems of the last gradient that took the sequential kernel
  double cert_resid = 1e-11;          // end-state mismatch of the chunked replay that still counts as consistent
  double cert_gamma = 1e7;            // conditioning record gamma_max / mu_min above which a problem leaves the replay-free route
  double cert_gamma_abs = 1e4;        // ... and gamma_max alone (decide_kernel; calibration: profiles/r03_conditioning_calibration.txt)
  double cert_eg = 3e-9;              // ... and gamma_max x the largest measured G error of the chunks
  int summarize_mode = -1;            // -1 auto, 0 single wave, 1 role split (widths 7, 8)
  // warm-started plain recurrence for series that forget their past (clr_batch_kernels.h: warm_kernel)
  int warm_mode = -1;                 // -1 auto (per problem, from the decay over the samples before the chunk boundaries),
                                      // 0 off, 1 every problem with warm_forced_K warm-up steps (tests: the boundary check decides)
  int warm_forced_K = 0;
  int warm_explicit_chunks = 0;       // chunk count asked for through clr_batch_set_chunks (0: automatic)
  int wnchunk = 0, wL = 0;            // the warm path's own chunking
  static const int WARM_NK = 10;
  int warm_cand[WARM_NK] = {8, 12, 16, 24, 32, 48, 64, 80, 96, 128};
  std::vector<double> warm_span;      // [B or 1][WARM_NK] shortest time the K samples before a chunk boundary span
  std::vector<int> warm_K;            // [B] warm-up steps per problem of the current coefficients (0: scan)
  std::vector<double> host_cmin;      // [B] slowest decay rate of problem b (last set_coefficients)
  bool warm_K_dirty = false;          // warm_K changed on the host after the last upload (set_series re-selected it)
  bool warm_active = false;           // the current (series, coefficients) pair runs the warm path
  bool warm_inflight = false;         // results of a warm evaluation have not been looked at yet
  bool small_inflight = false;        // ... of a one-launch evaluation (small_batch_kernel): pending problems, no warm statistics
  bool in_fallback = false;           // building the parameters of the scan behind the warm path
  int warm_boost = 0;                 // candidates skipped after an evaluation with many fallbacks
  int warm_clean = 0;                 // consecutive warm evaluations without a fallback (decays warm_boost)
  int warm_settled = 0, warm_fallbacks = 0;  // of the last evaluation
  DevBuf wstarts, wends, wpart, wresid;
  DevBuf wT, wD, wY;                  // the warm kernel's padded chunk-interleaved copy of the series
  int wKpad = 0, wrows = 0;
  bool warm_copy_pending = true;
  DevBuf sT, sD, sY;                  // the one-launch path's chunk-interleaved copy of the series (small_params, api_batch.hip)
  bool small_copy_pending = true;
  int* wints = nullptr;               // wflags [B * wnchunk] | need_scan [B] | K [B]
  size_t wints_cap = 0;
  // general terms for the whole batch (clr_batch_set_general): the plan then evaluates through the any-width sequential
  // kernel, one workgroup per problem (generic_kernels.hip: generic_loglike_batch_kernel)
  int J_general = 0;
  DevBuf gA, gU, gV;
  long gA_stride = 0, gU_stride = 0, gV_stride = 0;
  int replay_source = -1;             // where the replay reads the series when summarize reads the chunk-interleaved
                                      // copy: 0 the same copy, 1 the row-major arrays staged through LDS, -1 auto
  bool relayout_pending = true;
  bool have_series = false, have_coeffs = false, have_factor = false;
  bool evaluated = false;             // an evaluation has been enqueued since the plan was (re)chunked
  DevBuf elems, starts, part, partx, cond, out;  // out: ll | logdet | quad | status (B ints)
  int* flags = nullptr;                    // flags [B*nchunk] | flagsx [B*nchunk] | need_exact [B]
  int force_exact = 0;
  bool factor_valid = false;  // a materialising run has written the factor under the chunking in force
  DevBuf bs_decay, bs_y;                        // clr_batch_dot_L / clr_batch_dot: the chunks' decay products; dot's output, chunk-interleaved
  DevBuf bs_rm, bs_x, bs_M, bs_off, bs_starts;  // clr_batch_solve: right-hand sides row-major / chunk-interleaved, chunk maps, offsets, start states
  bool bs_M_valid = false;                      // bs_M holds the chunk maps of the factor in HBM (they depend on the factor only)
  double solve_device_ms = 0.0;                 // device time of the last clr_batch_solve (HIP events around its kernels)
  hipEvent_t bs_ev[2] = {nullptr, nullptr};     // ... its two timing events, created by the first solve, kept for the plan's life
  int factor_layout = 0;      // clr_batch_set_factor_layout: 0 the reference's four arrays, 1 lean (W, D; phi, u regenerated)
  bool factor_is_lean = false;  // what the factor in HBM holds (set by the materialising run that wrote it)
  bool factor_inputs_changed = false;  // series or coefficients replaced since that run (a lean factor can then no longer be expanded)
  DevBuf phi, u, W, D;        // materialised factor, chunk-interleaved device layout
  // heads of the chunks recomputed from the previous chunk's replayed end state (clr_batch_set_factor_refine; BatchParams::ends)
  int factor_refine = 64;     // samples per chunk head (0: off -- round 5's factor)
  DevBuf ends;                // [B][nchunk][START] states at the chunks' ends, written by the materialising replay
  DevBuf fphi, fu, fW, fD;    // one problem in the reference's storage (get_factor)
  // materialising runs as a pipeline over groups of problems (clr_batch_set_materialize_pipeline): the summarize of
  // group g + 1 (fp64-VALU-bound) runs beside the replay of group g (HBM-bound) on streams that own disjoint sets of CUs
  int mp_groups = 0, mp_cus = 0, mp_nstreams = 1;
  std::vector<hipStream_t> mp_s;        // summarize streams (CU-masked when mp_cus > 0)
  hipStream_t mp_p = nullptr, mp_r = nullptr;  // prefix + corrections (any CU); replay (the other CUs)
  std::vector<hipEvent_t> mp_ev;        // [0] start, [1 + 2 g] group g summarized, [2 + 2 g] its start states ready, [last] replay done
  // problems the conditioning record sends to the checked chunked replay (level 1), re-planned as a small plan of their
  // own with many short chunks instead of replaying long chunks sequentially beside an idle chip (clr_batch_set_rescue)
  int rescue_mode = -1;            // -1 auto (chunks of >= 1024 samples), 0 off: the inline chunked replay, 1 whenever possible
  bool rescue_inflight = false;    // the evaluation in flight deferred its level-1 problems (pending until resolved)
  bool is_rescue_plan = false;     // this plan IS such a side plan (never defers)
  struct clr_batch* rescue = nullptr;
  int* rescue_idx = nullptr;       // device: the re-planned problems' indices
  size_t rescue_idx_cap = 0;
  int rescue_last = 0;             // problems of the last resolved evaluation that were re-planned (or replayed inline: negative)
  long rescue_total = 0;
  long rescue_plan_key = -1;       // the batch-wide count of pending problems the side plan in `rescue` was chunked for
  // A plan that is ONE SLICE of a larger batch (csrc/sharded.cpp, clr_group_hooks.h): every decision that looks at a
  // count over "the plan's problems" -- the prefix plan's time model, the one-launch path, the warm path's activation
  // and adaptation, deferring level-1 problems, side plan or inline replay and the side plan's chunk count -- is
  // taken from the counts over the WHOLE batch, so that a problem's result does not depend on the sharding.
  int group_B = 0;                 // problems of the whole batch (0: this plan is the whole batch)
  long warm_eligible = 0;          // problems of this plan the warm path could start at the coefficients in force
  long warm_eligible_total = -1;   // ... of the whole batch, as handed down (group_B > 0 only)
  // warm_resolve in two halves (resolve_begin / resolve_finish): the state between them
  bool res_open = false, res_was_warm = false, res_was_rescue = false;
  long res_pending = 0;
  bool pin_results = false;        // the pinned staging buffer holds the final results of the evaluation in force
  // optional per-kernel HIP events around the launches of clr_batch_enqueue (clr_batch_set_profiling)
  int prof_on = 0, prof_steps = 0;
  std::vector<hipEvent_t> prof_events;  // 7 per recorded step
};



// ---- a plan's state as kernel parameters -----------------------------------------------------------------------
namespace {
// the maxima the kernel selection looks at: the plan's own, raised to the floors of a sharded parent
double sel_max(double own, double floor) {  // (NaN on either side wins: the conservative kernels)
  if (own != own) return own;
  if (floor != floor) return floor;
  return own >= floor ? own : floor;
}
// the batch size the selection rules look at: the whole batch's when the plan is a slice of one (clr_batch::group_B)
int sel_B(const clr_batch* h) { return h->group_B > 0 ? h->group_B : h->B; }
bool lazy_eligible(const clr_batch* h) {
  // |c dx| < 2^-7 at every step: Psi stays within [0.88, 1] over the 16 steps between renormalisations;
  // |d dx| < 2^-5: the per-step rotation of the (cos, sin) pairs uses a short Taylor series
  const double cmax = sel_max(h->cmax, h->floor_cmax), dmax = sel_max(h->dmax, h->floor_dmax),
               dxmax = sel_max(h->dxmax, h->floor_dxmax);
  return h->have_series && h->have_coeffs && cmax * dxmax < 0.0078125 && dmax * dxmax < 0.03125;
}

// Wide plans (wave per (problem, chunk), widths 9..64): the lazy flavour of the summarize takes any series since round 5 --
// a lane whose own interval is too long for the Taylor steps sends its wave through the full sincos / exp for that batch
// (wide_scan_body, feature_batch) -- as long as the decay accumulated between two renormalisations (64 steps) stays far
// from the exponent range: Psi^-2 < e^(2 x 64 x 2) = e^256.  (The one-wave kernels of CLR_WIDE64_ONE_WAVE keep the strict rule.)
bool lazy_eligible_wide(const clr_batch* h) {
  // (read per call, like the launcher's own getenv: a process may toggle them between plans)
  if (clr::option("CLR_WIDE64_ONE_WAVE") && h->J > 32) return lazy_eligible(h);
  const char* env = clr::option("CLR_WIDE_LAZY_BOUND");  // (tuning runs: 0 = the strict rule of the narrow kernels)
  const double bound = env ? atof(env) : 2.0;
  // widths 9..16 (four lanes per row): the two flavours cost the same there -- 6.5 against 6.7 ms on a dense series, 7.4
  // against 7.0 on one where EVERY batch takes the slow path (profiles/r05m_wide_lazy_gaps.txt) -- so they keep the strict rule
  if (!(bound > 0.0) || h->J + h->J_general <= 16) return lazy_eligible(h);
  const double cmax = sel_max(h->cmax, h->floor_cmax), dxmax = sel_max(h->dxmax, h->floor_dxmax);
  return h->have_series && h->have_coeffs && cmax * dxmax < bound;
}

bool split_active(const clr_batch* h) {
  // explicit modes 1 / 2, or auto (-1):
  //  * widths 7 and 8 on a densely sampled series: the split kernel with the decay factored out of the state
  //    (lazy) beats the single-wave kernel for every shape (2.0-2.5 ms against 2.3-3.6, profiles/r02zzz_split_ab_shapes.txt);
  //  * any other series: the plain split at width 7 (2.5-2.7 ms against 2.8-3.0 on the paper's sparse family) and at
  //    width 8 with at least two complex terms (3.1-3.2 against 3.8-4.2; profiles/r02zzz_sparse_ab.txt).  With
  //    fewer complex terms at width 8 its trajectory wave spills ((8,0), (6,1): 4.1-4.2 against 3.7): single wave.
  if (!(h->launch && h->nchunk > 1 && clr::have_summarize_split(h->J_real, h->J_comp))) return false;
  if (h->in_fallback) return false;  // the scan behind the warm path: single-wave kernels on the row-major arrays
  if (h->summarize_mode > 0) return true;
  if (h->summarize_mode < 0 && h->J >= 7 && lazy_eligible(h)) return true;
  return h->summarize_mode < 0 && (h->J == 7 || (h->J == 8 && h->J_comp >= 2));
}

// Level-1 problems (ill-conditioned, not flagged) are left pending and re-planned with short chunks instead of being
// replayed inline: when the plan's chunks are long enough that ONE such problem would cost the whole batch a sequential
// chunk-time (config 4: +11 ms on 12.7; profiles/r04zz_wide_midbatch.txt) -- a re-plan is three passes over chunks a
// tenth as long.  Never on forced-exact / materialising runs (they replay everything), inside the gradient's own
// evaluation, behind the warm path, or on a side plan.
bool defer_runs(const clr_batch* h, int materialize) {
  if (h->rescue_mode == 0 || h->is_rescue_plan || materialize || h->force_exact || h->grad_scan_only || h->in_fallback) return false;
  if (h->nchunk < 2 || h->J_general > 0 || sel_B(h) < 2) return false;
  if (h->rescue_mode == 1) return true;
  // widths 33..64: a side plan is itself <= 16 chunks chained by a walk of ~0.6 ms each -- it does not beat the inline
  // replay of one of the parent's chunks (profiles/r05d_wide64_chunks.txt)
  if (h->J > 32) return false;
  return h->L >= 1024;
}

int batch_params(clr_batch* h, int materialize, clr::BatchParams& P) {
  if (!h->have_series || !h->have_coeffs)
    return fail(CLR_INVALID_ARGUMENT, "set_series and set_coefficients must be called first");
  int st = CLR_OK;
  if (materialize && !h->have_factor) {
    // widths 1..8: chunk-interleaved device layout (replay_chunk, MATERIALIZE == 2); widths 9..64: the wide kernels
    // write the reference's own storage per problem (wide_scan_kernel, MODE 0: phi, u [N-1][J], W [N][J], D [N])
    const size_t B = (size_t)h->B, J = (size_t)h->J, cells = h->launch ? (size_t)h->L * h->nchunk : (size_t)h->N;
    const bool lean = h->launch && h->factor_layout == 1;  // (phi and u are not stored: 8 N (J + 1) instead of 8 N (3 J + 1) bytes per problem)
    if (!lean && (st = h->phi.reserve(B * J * cells)) != CLR_OK) return st;
    if (!lean && (st = h->u.reserve(B * J * cells)) != CLR_OK) return st;
    if ((st = h->W.reserve(B * J * cells)) != CLR_OK) return st;
    if ((st = h->D.reserve(B * cells)) != CLR_OK) return st;
    h->have_factor = true;
  }
  memset(&P, 0, sizeof(P));
  const size_t B = (size_t)h->B, nr = B * h->J_real, nc = B * h->J_comp;
  P.B = h->B; P.N = h->N; P.nchunk = h->nchunk; P.L = h->L; P.L0 = h->L0;
  P.fast_trig = (!h->force_library_trig &&
                 sel_max(h->dmax, h->floor_dmax) * sel_max(h->tmax, h->floor_tmax) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
  P.coop_prefix = h->coop_prefix;
  P.plan = h->plan;
  P.lvl_elems = h->lvl_elems.p;
  P.scan_ws = (!h->launch && h->scan_ws_doubles) ? h->lvl_elems.p : nullptr;
  P.lvl_starts = h->lvl_starts.p;
  P.jitter = h->coeffs.p + 2 * nr + 4 * nc;
  P.a_real = h->coeffs.p;
  P.c_real = P.a_real + nr;
  P.a_comp = P.c_real + nr;
  P.b_comp = P.a_comp + nc;
  P.c_comp = P.b_comp + nc;
  P.d_comp = P.c_comp + nc;
  // role-split summarize (two waves per SIMD, clr_split_kernels.h) for the widths whose element
  // does not fit one wave's registers; it reads the chunk-interleaved copy of the series
  const bool split = split_active(h);
  P.split = split ? 1 : 0;
  P.split_lazy = (split && h->summarize_mode != 1 && lazy_eligible(h)) ? 1 : 0;
  // wide plans: the lazy-decay flavour of the wide summarize on dense series (mode 0 / 1 switch it off)
  // (1: dense everywhere -- the strict rule; 2: only the relaxed one: the flavour with the per-batch range test)
  if (!h->launch && h->nchunk > 1 && (h->summarize_mode < 0 || h->summarize_mode == 2) && lazy_eligible_wide(h))
    P.split_lazy = lazy_eligible(h) ? 1 : 2;
  // (the wide kernels, the warm-started recurrence and the scan behind it read the row-major arrays)
  if (h->launch && (h->layout == 1 || split) && h->nchunk > 1 && !h->in_fallback) {
    const long cells = (long)h->nchunk * h->L;
    auto nsrc = [&](long sd) { return (size_t)(sd == 0 ? 1 : h->B); };
    if ((st = h->tT.reserve(nsrc(h->t_stride) * cells)) != CLR_OK) return st;
    if ((st = h->dT.reserve(nsrc(h->diag_stride) * cells)) != CLR_OK) return st;
    if ((st = h->yT.reserve(nsrc(h->y_stride) * cells)) != CLR_OK) return st;
    P.t = h->tT.p; P.diag = h->dT.p; P.y = h->yT.p;
    P.t_stride = h->t_stride ? cells : 0;
    P.diag_stride = h->diag_stride ? cells : 0;
    P.y_stride = h->y_stride ? cells : 0;
    P.lane_is = h->nchunk; P.lane_cs = 1;
    P.staged = 0;
  } else {
    P.t = h->t.p; P.diag = h->diag.p; P.y = h->y.p;
    P.t_stride = h->t_stride; P.diag_stride = h->diag_stride; P.y_stride = h->y_stride;
    P.lane_is = 1; P.lane_cs = h->L;
    P.staged = ((h->layout == 2 || h->in_fallback) && h->nchunk > 1) ? 1 : 0;
  }
  P.elems = h->elems.p; P.starts = h->starts.p; P.part = h->part.p; P.flags = h->flags;
  P.cond = h->cond.p;
  P.cert_gamma = h->cert_gamma;
  P.cert_gamma_abs = h->cert_gamma_abs;
  P.cert_eg = h->cert_eg;
  P.egerr = h->cond.p + (size_t)h->B * h->nchunk * 3;
  P.cert_resid = h->cert_resid;
  P.head_cap = clr::output_check_cap(); P.head_tol = clr::output_check_tol();  // (materialising wide plans: BatchParams::head_check)
  {
    const size_t pc = B * (size_t)h->nchunk;
    P.partx = h->partx.p; P.flagsx = h->flags + pc; P.need_exact = h->flags + 2 * pc;
    // a single chunk starts from the zero state: its replay IS the whole recurrence
    P.force_exact = (materialize || h->force_exact || h->nchunk < 2) ? 1 : 0;
  }
  P.out_ll = h->out.p; P.out_logdet = h->out.p + B; P.out_quad = h->out.p + 2 * B;
  P.out_status = reinterpret_cast<int*>(h->out.p + 3 * B);
  if (h->wints) {
    const size_t wpc = B * (size_t)h->wnchunk;
    P.wflags = h->wints; P.need_scan = h->wints + wpc; P.wK = h->wints + wpc + B;
    P.wL = h->wL; P.wnchunk = h->wnchunk;
    P.wstarts = h->wstarts.p; P.wends = h->wends.p; P.wpart = h->wpart.p; P.wresid = h->wresid.p;
    P.warm_resid = h->cert_resid;
    const long cells = (long)h->wrows * h->wnchunk;
    P.wt = h->wT.p; P.wdiag = h->wD.p; P.wy = h->wY.p;
    P.wt_stride = h->t_stride ? cells : 0; P.wdiag_stride = h->diag_stride ? cells : 0; P.wy_stride = h->y_stride ? cells : 0;
    P.wKpad = h->wKpad; P.wrows = h->wrows;
  }
  P.only_pending = h->in_fallback ? 1 : 0;
  P.defer_level1 = defer_runs(h, materialize) ? 1 : 0;
  P.wide_materialize = (materialize && !h->launch) ? 1 : 0;
  P.ends = nullptr; P.ends_alt = nullptr; P.ends_in = nullptr;
  P.fixup_steps = 0;
  P.refine_samples = 0;
  {  // (the rotation of the phases needs |d dx| < 2^-5 at every step; the decay is not involved)
    const double dmax = sel_max(h->dmax, h->floor_dmax), dxmax = sel_max(h->dxmax, h->floor_dxmax);
    P.dense = (h->have_series && h->have_coeffs && dmax * dxmax < 0.03125) ? 1 : 0;
  }
  if (materialize && h->nchunk > 1 && h->factor_refine > 0 && h->J_general == 0 && h->J <= clr::wide_max_width()) {
    size_t START = 0;
    if (h->launch) START = (size_t)h->launch->start_doubles;
    else { const size_t JP = (size_t)clr::wide_padded_width(h->J); START = JP * (JP + 1) / 2 + JP; }
    if ((st = h->ends.reserve((size_t)h->B * h->nchunk * START * (h->launch ? 1 : 2))) != CLR_OK) return st;
    P.ends = h->ends.p;
    if (!h->launch) P.ends_alt = h->ends.p + (size_t)h->B * h->nchunk * START;  // (wide plans: the output check's second buffer)
    // (the recurrence needs a few J samples to forget a start state: wide plans take the setting per 16 rows of state)
    P.refine_samples = h->launch ? h->factor_refine : h->factor_refine * (clr::wide_padded_width(h->J) / 16);
  }
  P.phi = h->phi.p; P.u = h->u.p; P.W = h->W.p; P.D = h->D.p;
  return CLR_OK;
}

// The replay's view of the series.  The role-split summarize reads the chunk-interleaved copy; the replay is free to
// read either that copy (one 512-B line per array and step per wave, but a second 2.4 GB stream competing with the
// factor's stores) or the row-major arrays through the LDS-staged tiles (round 1's path).
clr::BatchParams replay_view(const clr_batch* h, const clr::BatchParams& P, int materialize) {
  clr::BatchParams R = P;
  const int src = h->replay_source < 0 ? 0 : h->replay_source;  // (measured: profiles/r03a_prefix_ab.txt)
  if (src == 1 && !P.staged && P.lane_cs == 1 && h->nchunk > 1 && h->layout == 2) {
    R.t = h->t.p; R.diag = h->diag.p; R.y = h->y.p;
    R.t_stride = h->t_stride; R.diag_stride = h->diag_stride; R.y_stride = h->y_stride;
    R.lane_is = 1; R.lane_cs = h->L;
    R.staged = 1;
  }
  return R;
}

// Row-major API layout -> chunk-interleaved layout (3 tiled transposes).  Returns whether the copy
// was (re)built: `relayout_pending` may only be cleared then -- the need for the copy can appear later
// (a new coefficient draw can switch the summarize kernel) with the series unchanged.
bool batch_relayout(clr_batch* h) {
  if (!((h->layout == 1 || split_active(h)) && h->nchunk > 1)) return false;
  const long cells = (long)h->nchunk * h->L;
  struct { DevBuf* src; DevBuf* dst; long stride; int pad; } jobs[3] = {
      {&h->t, &h->tT, h->t_stride, 1}, {&h->diag, &h->dT, h->diag_stride, 2}, {&h->y, &h->yT, h->y_stride, 0}};
  for (auto& j : jobs)
    clr::launch_relayout(j.src->p, j.stride, j.dst->p, j.stride ? cells : 0, j.stride ? h->B : 1,
                         h->N, h->L, h->nchunk, j.pad, h->stream);
  return true;
}


// CLR_OK, or CLR_HIP_ERROR when a kernel of the flow could not be configured (nothing after it is launched: the later
// kernels would run on stale start states)
int wide_flow(clr::BatchParams& P, int J_real, int J_comp, hipStream_t stream, hipEvent_t* ev) {
  auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], stream); };
  const int JP = clr::wide_padded_width(J_real + 2 * J_comp + P.J_general);
  mark(1);
  if (P.nchunk > 1) clr::launch_wide_summarize(P, J_real, J_comp, stream);
  mark(2);
  // widths 33..64: the prefix and the corrections in one walk per problem (wide64_kernels.hip); CLR_WIDE_WALK=1 takes
  // that kernel at the padded width 32 too (cross-check of the two-kernel path; tests)
  const bool walk32 = clr::option("CLR_WIDE_WALK") != nullptr;
  if (JP == 64 || (JP == 32 && walk32 && !(P.scan_ws && P.coop_prefix == 2))) {
    if (clr::launch_wide_walk(P, JP, stream) != 0)
      return fail(CLR_HIP_ERROR, "the width-64 walk kernel needs 132 KB of dynamic LDS per workgroup: the device refused the attribute");
    mark(3);
    clr::launch_wide_decide(P, JP, stream);
  } else {
    clr::launch_wide_prefix(P, JP, stream);
    mark(3);
    clr::launch_wide_correct(P, JP, stream);
  }
  mark(4);
  // one chunk: the sweep itself; several: the chunked replay of forced runs and of the problems the
  // conditioning record marked (level 1), with its end states checked against the scan
  clr::launch_wide_loglike(P, J_real, J_comp, stream);
  if (P.nchunk > 1) {
    clr::launch_wide_check_replay(P, stream);
    if (P.head_cap > 0.0 && P.ends && P.ends_alt && P.wide_materialize && P.cond) {
      // end states off the scanned start states, but not by much (level 3): the chunks again, each from the previous
      // chunk's replayed end state, until what two consecutive replays wrote agrees (BatchParams::head_check); an even
      // number of attempts, so that P.ends holds the last (or last but one: equal to head_tol) end states for the fix-up
      int attempts = 4;
      if (const char* e = clr::option("CLR_OUTPUT_CHECK_ATTEMPTS")) attempts = std::max(1, std::min(atoi(e), 16));  // (tools/gpu_reference_family_factor.py)
      for (int k = 0; k < attempts; ++k) {
        clr::BatchParams F = P;
        F.fixup_steps = P.L + (P.L0 > P.L ? P.L0 - P.L : 0);
        F.head_check = (k + 1 == attempts) ? 2 : 1;
        F.ends_in = (k & 1) ? P.ends_alt : P.ends;
        F.ends = (k & 1) ? P.ends : P.ends_alt;
        clr::launch_wide_loglike(F, J_real, J_comp, stream);
        clr::launch_wide_head_decide(F, stream);
      }
    }
  }
  if (P.nchunk > 1 && P.ends && P.wide_materialize && P.refine_samples > 0) {
    clr::BatchParams F = P;  // the heads of the chunks again, from the previous chunk's replayed end state
    F.fixup_steps = P.refine_samples;
    clr::launch_wide_loglike(F, J_real, J_comp, stream);
  }
  if (P.nchunk > 1) {
    clr::launch_finalize(P, stream);
    clr::BatchParams S = P;  // the flagged problems, sequentially
    S.nchunk = 1; S.L = P.N; S.L0 = 0; S.seq_only = 1; S.force_exact = 1; S.ends = nullptr; S.fixup_steps = 0;
    clr::launch_wide_loglike(S, J_real, J_comp, stream);
  }
  mark(5);
  mark(6);
  return CLR_OK;
}
int wide_launch(clr_batch* h, clr::BatchParams& P, hipEvent_t* ev) {
  return wide_flow(P, h->J_real, h->J_comp, h->stream, ev);
}

const int PROF_NK = 6, PROF_MAX_STEPS = 4096;


// a plan with general terms on the wide kernels: its own chunking and workspace (clr_batch_set_general)
void general_wide_params(const clr_batch* h, const clr::BatchParams& P, clr::BatchParams& W) {
  W = P;
  const size_t pc = (size_t)h->B * h->gen_nchunk;
  W.nchunk = h->gen_nchunk; W.L = h->gen_L; W.L0 = h->gen_L0;
  W.t = h->t.p; W.diag = h->diag.p; W.y = h->y.p;
  W.t_stride = h->t_stride; W.diag_stride = h->diag_stride; W.y_stride = h->y_stride;
  W.lane_is = 1; W.lane_cs = W.L; W.staged = 0; W.split = 0; W.only_pending = 0;
  W.split_lazy = ((h->summarize_mode < 0 || h->summarize_mode == 2) && W.nchunk > 1 && lazy_eligible(h)) ? 1 : 0;  // (general terms: the strict rule)
  W.coop_prefix = h->coop_prefix == 2 ? 2 : 1;  // (2: the parallel prefix where its workspace exists, else the walk)
  W.J_general = h->J_general;
  W.gen_A = h->gA.p; W.gen_U = h->gU.p; W.gen_V = h->gV.p;
  W.gen_A_stride = h->gA_stride; W.gen_U_stride = h->gU_stride; W.gen_V_stride = h->gV_stride;
  W.elems = h->gen_elems.p; W.starts = h->gen_starts.p;
  W.scan_ws = h->gen_scan_ws_doubles ? h->gen_scan.p : nullptr;
  W.part = h->gen_part.p; W.partx = h->gen_part.p + pc * 2;
  W.cond = h->gen_cond.p; W.egerr = h->gen_cond.p + pc * 3;
  W.flags = h->gen_flags; W.flagsx = h->gen_flags + pc; W.need_exact = h->gen_flags + 2 * pc;
  W.force_exact = (h->force_exact || W.nchunk < 2) ? 1 : 0;
  W.wide_materialize = 0;
}

}  // namespace
