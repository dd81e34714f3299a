// celerite_amd/csrc/api.hip -- the C ABI of include/celerite_hip.h: handles, HBM
// residency, path selection and kernel launches.  The recurrences run on the device
// only (without a gfx950 device every compute entry fails with CLR_NO_DEVICE); the one
// piece of host arithmetic is the O(N) diagonal a_n = diag_n + sum(a) + jitter (+ A_n)
// that the general-terms / wide single-solver path forms before its upload.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/celerite_hip.h"
#include "clr_batch_kernels.h"
#include "clr_carma.h"
#include "clr_generic_kernels.h"
#include "clr_series_io.h"
#include "clr_small.h"
#include "clr_wide.h"

namespace clr {
const BatchLaunchers* batch_launchers_w1(int, int);
const BatchLaunchers* batch_launchers_w2(int, int);
const BatchLaunchers* batch_launchers_w3(int, int);
const BatchLaunchers* batch_launchers_w4(int, int);
const BatchLaunchers* batch_launchers_w5(int, int);
const BatchLaunchers* batch_launchers_w6(int, int);
const BatchLaunchers* batch_launchers_w7(int, int);
const BatchLaunchers* batch_launchers_w8(int, int);

const BatchLaunchers* find_batch_launchers(int JR, int JC) {
  switch (JR + 2 * JC) {
    case 1: return batch_launchers_w1(JR, JC);
    case 2: return batch_launchers_w2(JR, JC);
    case 3: return batch_launchers_w3(JR, JC);
    case 4: return batch_launchers_w4(JR, JC);
    case 5: return batch_launchers_w5(JR, JC);
    case 6: return batch_launchers_w6(JR, JC);
    case 7: return batch_launchers_w7(JR, JC);
    case 8: return batch_launchers_w8(JR, JC);
    default: return nullptr;
  }
}

// One wave per problem: lane l sums chunks l, l + 64, ... in order, then a fixed butterfly -- the same tree whatever the
// batch size or sharding, so results stay bit-identical across shard counts.  (One thread per problem walking all
// chunks took 40 us at 125 chunks: a fifth of BASELINE config 1's step.)
__global__ void __launch_bounds__(64) finalize_kernel(const BatchParams P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (P.only_pending && P.need_scan[b] == 0) return;  // (settled and written by the warm path)
  // replay-free sums unless the problem was marked for (or the run forces) the exact replay
  const bool exact = P.force_exact || P.need_exact[b] != 0;
  const double* part = exact ? P.partx : P.part;
  const int* flags = exact ? P.flagsx : P.flags;
  double ld = 0.0, qd = 0.0;
  int bad = 0;
  for (int c = lane; c < P.nchunk; c += 64) {
    ld += part[((long)b * P.nchunk + c) * 2 + 0];
    qd += part[((long)b * P.nchunk + c) * 2 + 1];
    bad |= flags[(long)b * P.nchunk + c];
  }
  ld = row_sum<1>(ld);
  qd = row_sum<1>(qd);
  bad = __any(bad) ? 1 : 0;
  if (lane != 0) return;
  if (bad) {  // celerite::linalg_exception (cholesky.h:176); quiet => -inf (celerite.py:205-208)
    P.out_status[b] = CLR_NOT_POSITIVE_DEFINITE;
    P.out_ll[b] = -INFINITY;
    P.out_logdet[b] = NAN;
    P.out_quad[b] = NAN;
    return;
  }
  P.out_status[b] = CLR_OK;
  P.out_logdet[b] = ld;
  P.out_quad[b] = qd;
  P.out_ll[b] = combine_loglike(ld, qd, P.N);
}

void launch_finalize(const BatchParams& P, hipStream_t s) {
  hipLaunchKernelGGL(finalize_kernel, dim3(P.B), dim3(64), 0, s, P);
}

// Tiled transpose through LDS: reads coalesced along i (the time axis), writes
// coalesced along the chunk axis.  Pure data movement: 8 B in + 8 B out per sample.
// Cells past the end of the series (the tail of the last chunk) are filled so that a reader may treat them as
// ordinary samples: pad_kind 1 repeats the series' last value (t: dx = 0), 2 writes 1e300 (the diagonal: 1 / D ~ 0),
// 0 writes zeros (y).  The lazy role-split summarize reads them unguarded (clr_split_kernels.h); every other reader
// masks them and never sees the values.
__global__ void __launch_bounds__(256) relayout_kernel(const double* __restrict__ src,
                                                       long src_stride, double* __restrict__ dst,
                                                       long dst_stride, int N, int L, int nchunk, int pad_kind) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z, i0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const double* in = src + (long)b * src_stride;
  double* out = dst + (long)b * dst_stride;
  const double pad = pad_kind == 1 ? in[N - 1] : (pad_kind == 2 ? 1e300 : 0.0);
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int c = c0 + r, i = i0 + threadIdx.x;
    const long n = (long)c * L + i;
    tile[r][threadIdx.x] = (c < nchunk && i < L && n < N) ? in[n] : pad;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = i0 + r, c = c0 + threadIdx.x;
    if (i < L && c < nchunk) out[(long)i * nchunk + c] = tile[threadIdx.x][r];
  }
}

// [i][j][chunk] (one problem of a materialising batch run) -> the reference's
// [n][j] storage with its index conventions (u one column earlier, cholesky.h:131-151).
__global__ void __launch_bounds__(256) deinterleave_factor_kernel(
    const double* __restrict__ phi_i, const double* __restrict__ u_i,
    const double* __restrict__ W_i, const double* __restrict__ D_i, double* __restrict__ phi,
    double* __restrict__ u, double* __restrict__ W, double* __restrict__ D, int N, int J, int L,
    int nchunk) {
  const long total = (long)N * J;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / J), j = (int)(e % J);
    const int c = n / L, i = n % L;
    const long src = ((long)i * J + j) * nchunk + c;
    W[e] = W_i[src];
    if (n + 1 < N) phi[e] = phi_i[src];
    if (n >= 1) u[e - J] = u_i[src];
    if (j == 0) D[n] = D_i[(long)i * nchunk + c];
  }
}

void launch_deinterleave_factor(const double* phi_i, const double* u_i, const double* W_i,
                                const double* D_i, double* phi, double* u, double* W, double* D,
                                int N, int J, int L, int nchunk, hipStream_t s) {
  const long total = (long)N * J;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(deinterleave_factor_kernel, dim3(blocks), dim3(256), 0, s, phi_i, u_i, W_i,
                     D_i, phi, u, W, D, N, J, L, nchunk);
}

// The warm kernel's copy: [row][chunk] per problem with row r of chunk c = sample n = c L - Kpad + r, rows =
// Kpad + L + 8 (every chunk's column starts with the Kpad samples in front of it -- its warm-up -- and ends with the
// look-ahead of its last steps).  Samples outside the series are padding a recurrence step ignores: before the
// start t = t_0, after the end t = t_{N-1} (a zero time step), the diagonal 1e300 (1 / D ~ 0), y = 0.
__global__ void __launch_bounds__(256) relayout_warm_kernel(const double* __restrict__ src, long src_stride,
                                                            double* __restrict__ dst, long dst_stride, int N, int L,
                                                            int nchunk, int Kpad, int rows, int pad_kind) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const double* in = src + (long)b * src_stride;
  double* out = dst + (long)b * dst_stride;
  const double pad_lo = pad_kind == 1 ? in[0] : (pad_kind == 2 ? 1e300 : 0.0);
  const double pad_hi = pad_kind == 1 ? in[N - 1] : (pad_kind == 2 ? 1e300 : 0.0);
  for (int q = threadIdx.y; q < 32; q += 8) {
    const int c = c0 + q, r = r0 + threadIdx.x;
    const long n = (long)c * L - Kpad + r;
    double v = n < 0 ? pad_lo : pad_hi;
    if (c < nchunk && r < rows && n >= 0 && n < N) v = in[n];
    tile[q][threadIdx.x] = v;
  }
  __syncthreads();
  for (int q = threadIdx.y; q < 32; q += 8) {
    const int r = r0 + q, c = c0 + threadIdx.x;
    if (r < rows && c < nchunk) out[(long)r * nchunk + c] = tile[threadIdx.x][q];
  }
}

void launch_relayout_warm(const double* src, long src_stride, double* dst, long dst_stride, int nsrc, int N, int L,
                          int nchunk, int Kpad, int rows, int pad_kind, hipStream_t s) {
  dim3 grid((rows + 31) / 32, (nchunk + 31) / 32, nsrc);
  hipLaunchKernelGGL(relayout_warm_kernel, grid, dim3(32, 8), 0, s, src, src_stride, dst, dst_stride, N, L, nchunk,
                     Kpad, rows, pad_kind);
}

void launch_relayout(const double* src, long src_stride, double* dst, long dst_stride, int nsrc,
                     int N, int L, int nchunk, int pad_kind, hipStream_t s) {
  dim3 grid((L + 31) / 32, (nchunk + 31) / 32, nsrc);
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(32, 8), 0, s, src, src_stride, dst, dst_stride,
                     N, L, nchunk, pad_kind);
}
}  // namespace clr

namespace {

thread_local std::string g_last_error;
thread_local int g_device = 0;

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
  DevBuf ws_elems, ws_starts, ws_part, ws_cond;  // scan workspace
  DevBuf ws_lvl_elems, ws_lvl_starts;            // upper levels of the multi-level prefix
  DevBuf gradbuf;                       // grad_log_likelihood staging
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
  DevBuf gen_elems, gen_starts, gen_part, gen_cond;
  int* gen_flags = nullptr;
  bool pipeline_pinned = false;    // the caller tuned the scan pipeline (chunks, prefix, summarize kernel, layout, certificate): auto small mode stays out
  double wide_first_ratio = 1.25;  // wide plans: cost of a chunk with riders / cost of the riderless first chunk
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
  bool in_fallback = false;           // building the parameters of the scan behind the warm path
  int warm_boost = 0;                 // candidates skipped after an evaluation with many fallbacks
  int warm_clean = 0;                 // consecutive warm evaluations without a fallback (decays warm_boost)
  int warm_settled = 0, warm_fallbacks = 0;  // of the last evaluation
  DevBuf wstarts, wends, wpart, wresid;
  DevBuf wT, wD, wY;                  // the warm kernel's padded chunk-interleaved copy of the series
  int wKpad = 0, wrows = 0;
  bool warm_copy_pending = true;
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
  DevBuf phi, u, W, D;        // materialised factor, chunk-interleaved device layout
  DevBuf fphi, fu, fW, fD;    // one problem in the reference's storage (get_factor)
  // optional per-kernel HIP events around the launches of clr_batch_enqueue (clr_batch_set_profiling)
  int prof_on = 0, prof_steps = 0;
  std::vector<hipEvent_t> prof_events;  // 7 per recorded step
};


namespace {

int ensure_stream(clr_solver* s) {
  int st = require_device(s->device);
  if (st != CLR_OK) return st;
  if (!s->have_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_status), sizeof(int) * 4));
    s->have_stream = true;
  }
  return CLR_OK;
}

clr::GenericProblem generic_view(const clr_solver* s) {
  clr::GenericProblem g;
  g.N = s->N;
  g.J = s->J;
  g.J_real = s->J_real;
  g.J_comp = s->J_comp;
  g.J_general = s->J_general;
  const double* c = s->coeffs.p;
  g.a_real = c;
  g.c_real = c + s->J_real;
  g.a_comp = c + 2 * s->J_real;
  g.b_comp = g.a_comp + s->J_comp;
  g.c_comp = g.b_comp + s->J_comp;
  g.d_comp = g.c_comp + s->J_comp;
  g.U = s->U.p;
  g.V = s->V.p;
  g.t = s->t.p;
  return g;
}

int reserve_flags(int*& p, size_t& cap, size_t n) {
  if (n <= cap && p) return CLR_OK;
  if (p) (void)hipFree(p);
  p = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(int)));
  cap = std::max<size_t>(n, 1);
  return CLR_OK;
}

int check_coeff_dims(int n_a_real, int n_c_real, int n_a_comp, int n_b_comp, int n_c_comp,
                     int n_d_comp, int n_A, int U_rows, int U_cols, int V_rows, int V_cols,
                     int N) {
  // cholesky.h:59-69 / :459-469
  if (n_a_real != n_c_real || n_a_comp != n_b_comp || n_a_comp != n_c_comp ||
      n_a_comp != n_d_comp)
    return CLR_DIMENSION_MISMATCH;
  const bool has_general = (n_A != 0);
  if (has_general && (n_A != N || U_cols != N || V_cols != N)) return CLR_DIMENSION_MISMATCH;
  if (U_rows != V_rows) return CLR_DIMENSION_MISMATCH;
  return CLR_OK;
}

// Packs the six coefficient blocks contiguously and uploads them.
int upload_coeffs(DevBuf& buf, int J_real, const double* a_real, const double* c_real,
                  int J_comp, const double* a_comp, const double* b_comp, const double* c_comp,
                  const double* d_comp, hipStream_t stream, std::vector<double>& host) {
  host.clear();
  host.insert(host.end(), a_real, a_real + J_real);
  host.insert(host.end(), c_real, c_real + J_real);
  host.insert(host.end(), a_comp, a_comp + J_comp);
  host.insert(host.end(), b_comp, b_comp + J_comp);
  host.insert(host.end(), c_comp, c_comp + J_comp);
  host.insert(host.end(), d_comp, d_comp + J_comp);
  return upload(buf, host.data(), host.size(), stream);
}

// compute's uploads: through the solver's pinned arena when they fit (reset at the start of every compute,
// which ends with a stream synchronisation: nothing is in flight when a slice is reused)
void arena_reset(clr_solver* s, size_t want_doubles) {
  s->pin_off = 0;
  const size_t LIMIT = (size_t)1 << 20;  // 8 MB of pinned memory per solver at most
  if (want_doubles > LIMIT) return;      // (long series: the copies are bandwidth-, not latency-bound)
  if (want_doubles > s->pin_cap) {
    if (s->pin) (void)hipHostFree(s->pin);
    s->pin = nullptr;
    s->pin_cap = 0;
    void* p = nullptr;
    if (hipHostMalloc(&p, want_doubles * sizeof(double), hipHostMallocDefault) == hipSuccess) {
      s->pin = static_cast<double*>(p);
      s->pin_cap = want_doubles;
    } else {
      (void)hipGetLastError();
    }
  }
}
double* arena_take(clr_solver* s, size_t n) {
  if (!s->pin || s->pin_off + n > s->pin_cap) return nullptr;
  double* p = s->pin + s->pin_off;
  s->pin_off += n;
  return p;
}
int stage_upload(clr_solver* s, DevBuf& buf, const double* host, size_t n) {
  double* p = arena_take(s, n);
  if (!p) return upload(buf, host, n, s->stream);
  memcpy(p, host, n * sizeof(double));
  return upload(buf, p, n, s->stream);
}

}  // namespace

extern "C" {

/* ---- library / device ------------------------------------------------------ */
const char* clr_version(void) { return "0.3.0"; }
const char* clr_last_error(void) { return g_last_error.c_str(); }

const char* clr_status_string(int status) {
  switch (status) {
    case CLR_OK: return "ok";
    case CLR_DIMENSION_MISMATCH: return "dimension mismatch";
    case CLR_NOT_POSITIVE_DEFINITE: return "failed to factorize or solve matrix";
    case CLR_NOT_COMPUTED: return "you must call 'compute' first";
    case CLR_NO_DEVICE: return "no gfx950 device available (libcelerite_hip has no CPU path)";
    case CLR_HIP_ERROR: return "HIP runtime error";
    case CLR_INVALID_ARGUMENT: return "invalid argument";
    case CLR_UNSUPPORTED: return "unsupported configuration";
    case CLR_CARMA_INSTABILITY: return "CARMA model encountered an instability";
    default: return "unknown status";
  }
}

int clr_device_count(void) { return visible_gfx950(); }

int clr_set_device(int device) {
  int st = require_device(device);
  if (st == CLR_OK) g_device = device;
  return st;
}

int clr_get_device(int* device) {
  *device = g_device;
  return CLR_OK;
}

int clr_device_synchronize(void) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  HIP_TRY(hipDeviceSynchronize());
  return CLR_OK;
}

int clr_device_info(char* name, size_t name_len, int* compute_units, size_t* hbm_bytes) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_device));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return CLR_OK;
}

int clr_device_memory(size_t* free_bytes, size_t* total_bytes) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return CLR_OK;
}

/* ---- single-problem solver --------------------------------------------------- */
clr_solver* clr_solver_create(void) {
  clr_solver* s = new clr_solver();
  s->device = g_device;
  return s;  // device resources are acquired lazily, so construction never fails
}

void clr_solver_destroy(clr_solver* s) {
  if (!s) return;
  if (s->grad_plan) clr_batch_destroy(s->grad_plan);
  if (s->have_stream) {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    for (DevBuf* b : {&s->phi, &s->u, &s->W, &s->D, &s->coeffs, &s->t, &s->U, &s->V,
                      &s->scratch, &s->scratch2, &s->scalars, &s->ws_elems, &s->ws_starts,
                      &s->ws_part, &s->ws_cond, &s->gradbuf, &s->rhs, &s->ws_lvl_elems, &s->ws_lvl_starts})
      b->release();
    if (s->ws_flags) (void)hipFree(s->ws_flags);
    if (s->d_status) (void)hipFree(s->d_status);
    if (s->pin) (void)hipHostFree(s->pin);
    (void)hipStreamDestroy(s->stream);
  }
  delete s;
}

static void wide_flow(clr::BatchParams& P, int J_real, int J_comp, hipStream_t stream, hipEvent_t* ev);

int clr_solver_compute(clr_solver* s, double jitter, int n_a_real, const double* a_real,
                       int n_c_real, const double* c_real, int n_a_comp, const double* a_comp,
                       int n_b_comp, const double* b_comp, int n_c_comp, const double* c_comp,
                       int n_d_comp, const double* d_comp, int n_A, const double* A, int U_rows,
                       int U_cols, const double* U, int V_rows, int V_cols, const double* V,
                       int n_x, const double* x, int n_diag, const double* diag) {
  const int N = n_x;
  s->computed = 0;  // cholesky.h:57
  s->have_quad = false;
  const bool use_rhs = s->rhs_hint && (int)s->host_rhs.size() == N;
  s->rhs_hint = false;  // (one shot)
  if (N != n_diag) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  int st = check_coeff_dims(n_a_real, n_c_real, n_a_comp, n_b_comp, n_c_comp, n_d_comp, n_A,
                            U_rows, U_cols, V_rows, V_cols, N);
  if (st != CLR_OK) return fail(st, "dimension mismatch");
  if (N < 1) return fail(CLR_INVALID_ARGUMENT, "compute needs at least one sample");
  const bool has_general = (n_A != 0);
  const int J_general = U_rows, J_real = n_a_real, J_comp = n_a_comp;
  const int J = J_real + 2 * J_comp + J_general;
  if (J > CLR_MAX_WIDTH) return fail(CLR_UNSUPPORTED, "width above CLR_MAX_WIDTH");
  // rows of U/V are only read when general terms are active (cholesky.h:148-152
  // would read them regardless; a non-empty U with empty A is a caller error)
  if (J_general > 0 && !has_general) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");

  st = ensure_stream(s);
  if (st != CLR_OK) return st;
  hipStream_t stream = s->stream;

  s->N = N;
  s->J = J;
  s->J_real = J_real;
  s->J_comp = J_comp;
  s->J_general = J_general;
  const size_t Nm1 = (size_t)(N - 1);
  if ((st = s->phi.reserve((size_t)J * Nm1)) != CLR_OK) return st;
  if ((st = s->u.reserve((size_t)J * Nm1)) != CLR_OK) return st;
  if ((st = s->W.reserve((size_t)J * N)) != CLR_OK) return st;
  if ((st = s->D.reserve((size_t)N)) != CLR_OK) return st;
  if ((st = s->scalars.reserve(8)) != CLR_OK) return st;

  HIP_TRY(hipStreamSynchronize(stream));  // (a previous upload may still read host_coeffs / the pinned arena)
  arena_reset(s, has_general ? 0 : (size_t)3 * N + 2 * J_real + 4 * J_comp + 16);
  {
    std::vector<double>& hc = s->host_coeffs;
    hc.clear();
    hc.insert(hc.end(), a_real, a_real + J_real);
    hc.insert(hc.end(), c_real, c_real + J_real);
    hc.insert(hc.end(), a_comp, a_comp + J_comp);
    hc.insert(hc.end(), b_comp, b_comp + J_comp);
    hc.insert(hc.end(), c_comp, c_comp + J_comp);
    hc.insert(hc.end(), d_comp, d_comp + J_comp);
  }
  s->coeffs_lazy = false;

  // One short series of a narrow kernel: the whole factorisation in ONE launch and one upload (small_kernels.hip);
  // it settles the problem itself when every chunk boundary is consistent and no pivot is flagged, and hands it to
  // the general route below otherwise.
  if (!has_general && clr::small_compute_supported(J_real, J_comp, N) && !getenv("CLR_NO_SMALL_SOLVER")) {
    const size_t ELEM = (size_t)J * J + 2 * J + (size_t)J * (J + 1);
    int threads = 64;
    while (threads < 256 && threads * 8 < N && (size_t)threads * 2 * ELEM * sizeof(double) <= 60000) threads *= 2;
    clr::SmallParams S;
    memset(&S, 0, sizeof(S));
    S.N = N;
    S.L = (N + threads - 1) / threads;
    memcpy(S.coeff, s->host_coeffs.data(), s->host_coeffs.size() * sizeof(double));
    S.jitter = jitter;
    // t | diag | right-hand side: one block of the pinned arena, one copy; t stays at the head of s->t (predict)
    const size_t words = (size_t)N * (use_rhs ? 3 : 2);
    if ((st = s->t.reserve((size_t)3 * N)) != CLR_OK) return st;
    double* stage = arena_take(s, words);
    if (stage) {
      memcpy(stage, x, (size_t)N * sizeof(double));
      memcpy(stage + N, diag, (size_t)N * sizeof(double));
      if (use_rhs) memcpy(stage + 2 * (size_t)N, s->host_rhs.data(), (size_t)N * sizeof(double));
      HIP_TRY(hipMemcpyAsync(s->t.p, stage, words * sizeof(double), hipMemcpyHostToDevice, stream));
      S.t = s->t.p; S.diag = s->t.p + N; S.y = use_rhs ? s->t.p + 2 * (size_t)N : nullptr;
      S.phi = s->phi.p; S.u = s->u.p; S.W = s->W.p; S.D = s->D.p;
      S.out = s->scalars.p;
      S.max_residual = 1e-11;
      double dmax = 0.0;
      for (int j = 0; j < J_comp; ++j) { const double m = fabs(d_comp[j]); if (!(m <= dmax)) dmax = m; }
      const bool fast = dmax * max_abs(x, N) < CLR_FAST_TRIG_LIMIT;
      if (clr::launch_small_compute(J_real, J_comp, S, threads, fast, stream)) {
        HIP_TRY(hipGetLastError());
        double back_local[4];
        double* pinned_back = arena_take(s, 4);
        double* back = pinned_back ? pinned_back : back_local;
        HIP_TRY(hipMemcpyAsync(back, s->scalars.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (back[0] == 0.0) {
          s->coeffs_lazy = true;
          s->log_det = back[1];
          if (use_rhs) { s->cached_quad = back[2]; s->have_quad = true; }
          s->computed = 1;
          return CLR_OK;
        }
        // (not settled: indefinite, ill-conditioned or inconsistent -- the general route decides)
        arena_reset(s, (size_t)3 * N + 2 * J_real + 4 * J_comp + 16);
      }
    }
  }
  if ((st = stage_upload(s, s->coeffs, s->host_coeffs.data(), s->host_coeffs.size())) != CLR_OK) return st;
  if ((st = stage_upload(s, s->t, x, (size_t)N)) != CLR_OK) return st;

  int h_status = 0;
  double h_logdet = 0.0;

  if (J == 0) {  // cholesky.h:90-95
    if ((st = upload(s->scratch, diag, (size_t)N, stream)) != CLR_OK) return st;
    clr::launch_diag_only(N, s->scratch.p, jitter, s->D.p, s->scalars.p, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h_logdet, s->scalars.p, sizeof(double), hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipStreamSynchronize(stream));
  } else if (!has_general && J <= 8 && clr::find_batch_launchers(J_real, J_comp)) {
    // fixed-width chunked scan, materialising the reference-layout factor
    const clr::BatchLaunchers* L = clr::find_batch_launchers(J_real, J_comp);
    if ((st = stage_upload(s, s->scratch, diag, (size_t)N)) != CLR_OK) return st;
    if ((st = stage_upload(s, s->scratch2, &jitter, 1)) != CLR_OK) return st;
    clr::BatchParams P;
    memset(&P, 0, sizeof(P));
    P.B = 1;
    P.N = N;
    {
      double dmax = 0.0;
      for (int j = 0; j < J_comp; ++j) { const double m = fabs(d_comp[j]); if (!(m <= dmax)) dmax = m; }
      P.fast_trig = (dmax * max_abs(x, N) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
    }
    P.nchunk = auto_chunks(1, N, J, true);
    P.L = (N + P.nchunk - 1) / P.nchunk;
    if (P.nchunk > 1 && P.L > 8) P.L = (P.L + 7) & ~7;  // 64-B aligned chunk rows for the tile loads
    P.nchunk = (N + P.L - 1) / P.L;  // drop empty trailing chunks
    // one problem: the whole chip is idle during the prefix, so the chunk elements are composed level by level
    // (clr_prefix_kernels.h) instead of walked one by one
    P.coop_prefix = 2;
    P.plan = clr::plan_prefix(P.nchunk, -1, 0, 1, J);
    {
      size_t le = 0, ls = 0;
      clr::multilevel_workspace(P.plan, J, &le, &ls);
      if (le && (st = s->ws_lvl_elems.reserve(le)) != CLR_OK) return st;
      if (ls && (st = s->ws_lvl_starts.reserve(ls)) != CLR_OK) return st;
      P.lvl_elems = s->ws_lvl_elems.p;
      P.lvl_starts = s->ws_lvl_starts.p;
    }
    if ((st = s->ws_elems.reserve((size_t)P.nchunk * L->elem_doubles)) != CLR_OK) return st;
    if ((st = s->ws_starts.reserve((size_t)P.nchunk * L->start_doubles)) != CLR_OK) return st;
    if ((st = s->ws_part.reserve((size_t)P.nchunk * 2)) != CLR_OK) return st;
    if ((st = s->ws_cond.reserve((size_t)P.nchunk * 4)) != CLR_OK) return st;
    if ((st = reserve_flags(s->ws_flags, s->ws_flags_cap, (size_t)P.nchunk + 1)) != CLR_OK) return st;
    const clr::GenericProblem g = generic_view(s);
    P.jitter = s->scratch2.p;
    P.a_real = g.a_real; P.c_real = g.c_real;
    P.a_comp = g.a_comp; P.b_comp = g.b_comp; P.c_comp = g.c_comp; P.d_comp = g.d_comp;
    // row-major arrays; with more than one chunk the kernels stage them through LDS
    if (use_rhs && (st = stage_upload(s, s->rhs, s->host_rhs.data(), (size_t)N)) != CLR_OK) return st;
    P.t = s->t.p; P.diag = s->scratch.p; P.y = use_rhs ? s->rhs.p : s->t.p;  // (without a hinted rhs y is irrelevant)
    P.lane_is = 1; P.lane_cs = P.L;
    P.staged = P.nchunk > 1 ? 1 : 0;
    P.elems = s->ws_elems.p; P.starts = s->ws_starts.p; P.part = s->ws_part.p;
    P.flags = s->ws_flags;
    // the factor is wanted: always the exact replay, which overwrites the zero-start sums
    P.partx = P.part; P.flagsx = P.flags; P.need_exact = s->ws_flags + P.nchunk; P.force_exact = 1;
    P.out_ll = s->scalars.p; P.out_logdet = s->scalars.p + 1; P.out_quad = s->scalars.p + 2;
    P.out_status = reinterpret_cast<int*>(s->scalars.p + 3);
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    P.cond = s->ws_cond.p; P.cert_gamma = 1e7; P.cert_gamma_abs = 1e4; P.cert_eg = 3e-9; P.egerr = s->ws_cond.p + (size_t)P.nchunk * 3; P.cert_resid = 1e-11; P.logdet_only = use_rhs ? 0 : 1;
    if (P.nchunk < 2) HIP_TRY(hipMemsetAsync(P.need_exact, 0, sizeof(int), stream));  // (no prefix kernel clears it)
    L->summarize(P, stream);
    L->prefix(P, stream);
    L->correct(P, stream);        // flags + conditioning record (its sums are overwritten by the replay)
    L->replay(P, 1, stream);      // chunked, from the scanned start states
    L->sequential(P, 1, stream);  // the whole recurrence in one lane if those cannot be trusted
    clr::launch_finalize(P, stream);
    HIP_TRY(hipGetLastError());
    double back_local[4];  // ll | logdet | quad | status (int in the 4th slot): one copy
    double* pinned_back = arena_take(s, 4);
    double* back = pinned_back ? pinned_back : back_local;
    HIP_TRY(hipMemcpyAsync(back, s->scalars.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    h_logdet = back[1];
    memcpy(&h_status, &back[3], sizeof(int));
    h_status = (h_status == CLR_NOT_POSITIVE_DEFINITE) ? 1 : 0;
    if (use_rhs && !h_status) { s->cached_quad = back[2]; s->have_quad = true; }
  } else if (!has_general && J >= 9 && J <= clr::wide_max_width()) {
    // widths 9..64 without general terms: the batched wide kernels on one problem -- one wave per
    // chunk with S distributed over the lanes, up to 16 chunks chained by the scan (widths <= 32),
    // the replay writing the factor in the reference's storage (instead of factor_generic_kernel:
    // one workgroup, five barriers per step)
    if ((st = stage_upload(s, s->scratch, diag, (size_t)N)) != CLR_OK) return st;
    if ((st = stage_upload(s, s->scratch2, &jitter, 1)) != CLR_OK) return st;
    clr::BatchParams P;
    memset(&P, 0, sizeof(P));
    P.B = 1;
    P.N = N;
    {
      double dmax = 0.0;
      for (int j = 0; j < J_comp; ++j) { const double m = fabs(d_comp[j]); if (!(m <= dmax)) dmax = m; }
      P.fast_trig = (dmax * max_abs(x, N) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
    }
    int nchunk = 1;
    if (J <= clr::wide_scan_max_width()) {
      // one problem: the chunk waves run side by side (t = a N / nchunk), the prefix walks the chunks
      // (t = p nchunk): nchunk = sqrt(a N / p).  Measured (profiles/r02y_single_wide_chunks.txt): a = 1.4 us per
      // sample, p = 15 us per chunk up to width 16; a = 2.0 us, p = 97 us (three 32^3 products on the matrix
      // cores + a Gauss-Jordan) above.
      nchunk = (int)lround(sqrt((double)N * (J <= 16 ? 0.096 : 0.0208)));
      if (nchunk > N / 256) nchunk = N / 256;
      if (nchunk < 2 || N < 2048) nchunk = 1;  // (short series: the six launches of the chunked flow cost more)
    }
    P.L = (N + nchunk - 1) / nchunk;
    P.nchunk = (N + P.L - 1) / P.L;
    const size_t pc = (size_t)P.nchunk, JP = J <= 16 ? 16 : 32, SZP = JP * (JP + 1) / 2;
    if ((st = s->ws_elems.reserve(pc * (JP * JP + JP + SZP + JP + SZP))) != CLR_OK) return st;
    if ((st = s->ws_starts.reserve(pc * (SZP + JP))) != CLR_OK) return st;
    if ((st = s->ws_part.reserve(pc * 4)) != CLR_OK) return st;
    if ((st = s->ws_cond.reserve(pc * 4)) != CLR_OK) return st;
    if ((st = reserve_flags(s->ws_flags, s->ws_flags_cap, 2 * pc + 1)) != CLR_OK) return st;
    const clr::GenericProblem g = generic_view(s);
    P.jitter = s->scratch2.p;
    P.a_real = g.a_real; P.c_real = g.c_real;
    P.a_comp = g.a_comp; P.b_comp = g.b_comp; P.c_comp = g.c_comp; P.d_comp = g.d_comp;
    if (use_rhs && (st = stage_upload(s, s->rhs, s->host_rhs.data(), (size_t)N)) != CLR_OK) return st;
    P.t = s->t.p; P.diag = s->scratch.p; P.y = use_rhs ? s->rhs.p : s->t.p;  // (without a hinted rhs y is irrelevant)
    P.lane_is = 1; P.lane_cs = P.L;
    P.elems = s->ws_elems.p; P.starts = s->ws_starts.p;
    P.part = s->ws_part.p; P.partx = s->ws_part.p + pc * 2;
    P.flags = s->ws_flags; P.flagsx = s->ws_flags + pc; P.need_exact = s->ws_flags + 2 * pc;
    P.cond = s->ws_cond.p; P.cert_gamma = 1e7; P.cert_gamma_abs = 1e4; P.cert_eg = 3e-9; P.egerr = s->ws_cond.p + (size_t)P.nchunk * 3; P.cert_resid = 1e-11; P.logdet_only = use_rhs ? 0 : 1;
    P.force_exact = 1;       // the factor is wanted: every chunk is replayed (and checked against the scan)
    P.wide_materialize = 1;
    P.coop_prefix = 1;
    P.out_ll = s->scalars.p; P.out_logdet = s->scalars.p + 1; P.out_quad = s->scalars.p + 2;
    P.out_status = reinterpret_cast<int*>(s->scalars.p + 3);
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    HIP_TRY(hipMemsetAsync(P.need_exact, 0, sizeof(int), stream));  // (one chunk: no prefix kernel clears it)
    wide_flow(P, J_real, J_comp, stream, nullptr);
    HIP_TRY(hipGetLastError());
    double back_local[4];
    double* pinned_back = arena_take(s, 4);
    double* back = pinned_back ? pinned_back : back_local;
    HIP_TRY(hipMemcpyAsync(back, s->scalars.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    h_logdet = back[1];
    memcpy(&h_status, &back[3], sizeof(int));
    h_status = (h_status == CLR_NOT_POSITIVE_DEFINITE) ? 1 : 0;
    if (use_rhs && !h_status) { s->cached_quad = back[2]; s->have_quad = true; }
  } else {
    // general terms (or widths above 64): diagonal summed on the host in the reference's
    // order (cholesky.h:98-99), recurrence on the device
    double sum_ar = 0.0, sum_ac = 0.0;
    for (int j = 0; j < J_real; ++j) sum_ar += a_real[j];
    for (int j = 0; j < J_comp; ++j) sum_ac += a_comp[j];
    std::vector<double> d0((size_t)N);
    for (int n = 0; n < N; ++n) {
      d0[n] = ((diag[n] + sum_ar) + sum_ac) + jitter;
      if (has_general) d0[n] += A[n];
    }
    if ((st = upload(s->D, d0.data(), (size_t)N, stream)) != CLR_OK) return st;
    if (J_general) {
      if ((st = upload(s->U, U, (size_t)J_general * N, stream)) != CLR_OK) return st;
      if ((st = upload(s->V, V, (size_t)J_general * N, stream)) != CLR_OK) return st;
    }
    const clr::GenericProblem g = generic_view(s);
    clr::launch_factor_generic(g, s->phi.p, s->u.p, s->W.p, s->D.p, s->d_status, s->scalars.p,
                               stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h_status, s->d_status, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&h_logdet, s->scalars.p, sizeof(double), hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipStreamSynchronize(stream));
  }

  if (h_status != 0)
    return fail(CLR_NOT_POSITIVE_DEFINITE, "failed to factorize or solve matrix");
  s->log_det = h_logdet;
  s->computed = 1;
  return CLR_OK;
}

int clr_solver_grad_log_likelihood(clr_solver* s, double jitter, int n_a_real, const double* a_real,
                                   int n_c_real, const double* c_real, int n_a_comp,
                                   const double* a_comp, int n_b_comp, const double* b_comp,
                                   int n_c_comp, const double* c_comp, int n_d_comp,
                                   const double* d_comp, int n_A, const double* A, int U_rows,
                                   int U_cols, const double* U, int V_rows, int V_cols,
                                   const double* V, int n_x, const double* x, int n_y,
                                   const double* y, int n_diag, const double* diag, double* value,
                                   int n_grad, double* grad) {
  const int N = n_x;
  if (N != n_diag || N != n_y) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  int st = check_coeff_dims(n_a_real, n_c_real, n_a_comp, n_b_comp, n_c_comp, n_d_comp, n_A,
                            U_rows, U_cols, V_rows, V_cols, N);
  if (st != CLR_OK) return fail(st, "dimension mismatch");
  if (N < 1) return fail(CLR_INVALID_ARGUMENT, "grad_log_likelihood needs at least one sample");
  const bool has_general = (n_A != 0);
  const int JG = U_rows, JR = n_a_real, JC = n_a_comp;
  if (JG > 0 && !has_general) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  if (JR + 2 * JC + JG > 64) return fail(CLR_UNSUPPORTED, "grad_log_likelihood supports widths up to 64");
  const int G = 1 + 2 * JR + 4 * JC;
  if (n_grad != G || !value || !grad) return fail(CLR_INVALID_ARGUMENT, "grad must hold 1 + 2 J_real + 4 J_comp values");
  if ((st = ensure_stream(s)) != CLR_OK) return st;
  hipStream_t stream = s->stream;

  const int Wc = JR + 2 * JC, Wt = Wc + JG;
  const bool narrow_plan = !has_general && JG == 0 && Wc >= 1 && Wc <= 8 && N >= 1024;
  // widths 9..32, and general terms up to a total width of 32: the wide scan + chunk-wise forward-mode tangents
  // (wide_batch_grad); chunk count for ONE problem from profiles/r04k_wide_grad_chunks.txt
  const bool wide_plan = !narrow_plan && Wc >= 1 && Wt <= 32 && (Wc >= 9 || JG > 0) && N >= 4096;
  if ((narrow_plan || wide_plan) && !getenv("CLR_GRAD_SEQUENTIAL")) {
    // parallel in n: the scan + the chunk-wise tangents (clr_batch_grad) on a one-problem plan
    if (!s->grad_plan || s->grad_N != N || s->grad_JR != JR || s->grad_JC != JC || s->grad_wide != wide_plan) {
      if (s->grad_plan) clr_batch_destroy(s->grad_plan);
      s->grad_plan = clr_batch_create(1, N, JR, JC, s->device);
      s->grad_series.clear();
      s->grad_N = N; s->grad_JR = JR; s->grad_JC = JC; s->grad_wide = wide_plan;
      if (s->grad_plan && wide_plan) {
        int nc = (int)lround(sqrt((double)N / (Wt <= 16 ? 40.0 : 170.0)));
        nc = std::max(Wt <= 16 ? 8 : 16, std::min(nc, Wt <= 16 ? 64 : 32));
        if ((st = clr_batch_set_chunks(s->grad_plan, nc)) != CLR_OK) return st;
      }
    }
    if (s->grad_plan) {
      const size_t n = (size_t)N;
      const bool same = s->grad_series.size() == 3 * n && !memcmp(s->grad_series.data(), x, n * sizeof(double)) &&
                        !memcmp(s->grad_series.data() + n, diag, n * sizeof(double)) &&
                        !memcmp(s->grad_series.data() + 2 * n, y, n * sizeof(double));
      if (!same) {
        if ((st = clr_batch_set_series(s->grad_plan, x, 0, diag, 0, y, 0)) != CLR_OK) return st;
        s->grad_series.resize(3 * n);
        memcpy(s->grad_series.data(), x, n * sizeof(double));
        memcpy(s->grad_series.data() + n, diag, n * sizeof(double));
        memcpy(s->grad_series.data() + 2 * n, y, n * sizeof(double));
      }
      if (wide_plan && (JG > 0 || s->grad_had_general)) {  // (general terms are arguments of every call)
        if ((st = clr_batch_set_general(s->grad_plan, JG, A, 0, U, 0, V, 0)) != CLR_OK) return st;
        s->grad_had_general = JG > 0;
      }
      if ((st = clr_batch_set_coefficients(s->grad_plan, &jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp)) != CLR_OK)
        return st;
      int pst = CLR_OK;
      if ((st = clr_batch_grad(s->grad_plan, value, grad, &pst)) != CLR_OK) return st;
      if (pst != CLR_OK) return fail(CLR_NOT_POSITIVE_DEFINITE, "failed to factorize or solve matrix");
      return CLR_OK;
    }
  }

  // one staging buffer: coefficients | A | U | V | t | diag | y | value, grad | status
  std::vector<double> host;
  host.reserve((size_t)2 * JR + 4 * JC + (size_t)N * (4 + 2 * JG));
  auto put = [&](const double* p, size_t n) { const size_t at = host.size(); if (n) host.insert(host.end(), p, p + n); return at; };
  const size_t o_ar = put(a_real, JR), o_cr = put(c_real, JR), o_ac = put(a_comp, JC), o_bc = put(b_comp, JC),
               o_cc = put(c_comp, JC), o_dc = put(d_comp, JC);
  const size_t o_A = put(A, has_general ? (size_t)N : 0), o_U = put(U, (size_t)JG * N), o_V = put(V, (size_t)JG * N);
  const size_t o_t = put(x, N), o_d = put(diag, N), o_y = put(y, N);
  const size_t o_out = host.size();
  if ((st = s->gradbuf.reserve(o_out + (size_t)G + 2)) != CLR_OK) return st;
  HIP_TRY(hipMemcpyAsync(s->gradbuf.p, host.data(), o_out * sizeof(double), hipMemcpyHostToDevice, stream));

  clr::GradParams P;
  memset(&P, 0, sizeof(P));
  const double* base = s->gradbuf.p;
  P.N = N; P.J_real = JR; P.J_comp = JC; P.J_general = JG;
  P.a_real = base + o_ar; P.c_real = base + o_cr; P.a_comp = base + o_ac; P.b_comp = base + o_bc;
  P.c_comp = base + o_cc; P.d_comp = base + o_dc;
  P.jitter = jitter;
  P.A = has_general ? base + o_A : nullptr; P.U = base + o_U; P.V = base + o_V;
  P.t = base + o_t; P.diag = base + o_d; P.y = base + o_y;
  {
    double dmax = 0.0;
    for (int j = 0; j < JC; ++j) { const double m = fabs(d_comp[j]); if (!(m <= dmax)) dmax = m; }
    P.fast_trig = (dmax * max_abs(x, N) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
  }
  P.out_value = s->gradbuf.p + o_out;
  P.out_grad = s->gradbuf.p + o_out + 1;
  P.out_status = s->d_status;
  clr::launch_grad(P, stream);
  HIP_TRY(hipGetLastError());
  std::vector<double> out((size_t)G + 1);
  int h_status = 0;
  HIP_TRY(hipMemcpyAsync(out.data(), s->gradbuf.p + o_out, out.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(&h_status, s->d_status, sizeof(int), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (h_status != CLR_OK) return fail(CLR_NOT_POSITIVE_DEFINITE, "failed to factorize or solve matrix");
  *value = out[0];
  for (int i = 0; i < G; ++i) grad[i] = out[(size_t)i + 1];
  if (!(jitter > 2.220446049250313e-16)) grad[0] = 0.0;  // solver.cpp:379-389,419-426
  return CLR_OK;
}

int clr_solver_hint_rhs(clr_solver* s, int n_b, const double* b) {
  if (n_b < 0 || (n_b > 0 && !b)) return fail(CLR_INVALID_ARGUMENT, "bad right-hand side");
  s->host_rhs.assign(b, b + n_b);
  s->rhs_hint = true;
  return CLR_OK;
}

int clr_solver_computed(const clr_solver* s) { return s->computed; }

int clr_solver_log_determinant(const clr_solver* s, double* out) {
  if (!s->computed) return fail(CLR_NOT_COMPUTED, "you must call 'compute' first");
  *out = s->log_det;
  return CLR_OK;
}

// dot_solve / solve as chunked scans for long series: N >= 2048 and width <= 32 one wave per chunk, one lane
// per column of the chunk's map (wsweep_kernels.hip); 256 <= N < 2048 and width <= 8 one lane per chunk
// (sweep_kernels.hip); otherwise the sequential sweeps (generic_kernels.hip)
static bool sweep_scan_ok(const clr_solver* s) {
  return clr::sweep_scan_supported(s->N, s->J) || clr::wsweep_scan_supported(s->N, s->J);
}
static int sweep_scan(clr_solver* s, int nrhs, const double* in, double* out, double* quad, int backward) {
  const bool wide = clr::wsweep_scan_supported(s->N, s->J);
  const int SLICE = 16384;  // right-hand sides per launch (grid.y / workspace bound); stream order keeps the slices apart
  for (int r0 = 0; r0 < nrhs; r0 += SLICE) {
    const int nr = std::min(SLICE, nrhs - r0);
    clr::SweepParams P;
    memset(&P, 0, sizeof(P));
    P.N = s->N; P.J = s->J; P.nrhs = nr;
    P.nchunk = wide ? clr::wsweep_chunks(s->N) : clr::sweep_chunks(s->N);
    P.L = (s->N - 1 + P.nchunk - 1) / P.nchunk;
    P.nchunk = (s->N - 1 + P.L - 1) / P.L;
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    P.in = in + (size_t)r0 * s->N;
    P.out = out ? out + (size_t)r0 * s->N : nullptr;
    P.quad = quad ? quad + r0 : nullptr;
    P.backward = backward;
    int st = s->ws_elems.reserve(wide ? clr::wsweep_workspace_doubles(s->J, P.nchunk, nr)
                                      : clr::sweep_workspace_doubles(s->J, P.nchunk, nr));
    if (st != CLR_OK) return st;
    if (wide) clr::launch_wsweep_scan(P, s->ws_elems.p, s->stream);
    else clr::launch_sweep_scan(P, s->ws_elems.p, s->stream);
  }
  return CLR_OK;
}

int clr_solver_dot_solve(const clr_solver* cs, int n_b, const double* b, double* out) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  if (n_b != s->N) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");  // :327
  if (!s->computed) return fail(CLR_NOT_COMPUTED, "you must call 'compute' first");
  // the vector hinted before compute: its quadratic form came out of compute's own pass
  if (s->have_quad && (int)s->host_rhs.size() == n_b &&
      memcmp(s->host_rhs.data(), b, sizeof(double) * (size_t)n_b) == 0) {
    *out = s->cached_quad;
    return CLR_OK;
  }
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  if ((st = upload(s->scratch, b, (size_t)s->N, s->stream)) != CLR_OK) return st;
  if ((st = s->scalars.reserve(8)) != CLR_OK) return st;
  if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, 1, s->scratch.p, nullptr, s->scalars.p, 0)) != CLR_OK) return st;
  } else {
    clr::launch_dot_solve(s->N, s->J, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                          s->scalars.p, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, s->scalars.p, sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return CLR_OK;
}

static int sweep_common(clr_solver* s, int rows, int nrhs, const double* in) {
  if (rows != s->N) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  if (!s->computed) return fail(CLR_NOT_COMPUTED, "you must call 'compute' first");
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  const size_t n = (size_t)s->N * (size_t)std::max(nrhs, 0);
  if ((st = upload(s->scratch, in, n, s->stream)) != CLR_OK) return st;
  return s->scratch2.reserve(n);
}

int clr_solver_solve(const clr_solver* cs, int b_rows, int nrhs, const double* b, double* x) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  int st = sweep_common(s, b_rows, nrhs, b);
  if (st != CLR_OK) return st;
  if (nrhs <= 0) return CLR_OK;
  if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, nrhs, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;   // :240-248
    if ((st = sweep_scan(s, nrhs, s->scratch2.p, s->scratch2.p, nullptr, 1)) != CLR_OK) return st;  // :249-259
  } else {
    clr::launch_solve(s->N, s->J, nrhs, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                      s->scratch2.p, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(x, s->scratch2.p, sizeof(double) * (size_t)s->N * nrhs,
                         hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return CLR_OK;
}

int clr_solver_dot_L(const clr_solver* cs, int z_rows, int nrhs, const double* z, double* y) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  int st = sweep_common(s, z_rows, nrhs, z);
  if (st != CLR_OK) return st;
  if (nrhs <= 0) return CLR_OK;
  const bool wide = clr::wdotl_scan_supported(s->N, s->J);
  if (wide || clr::sweep_scan_supported(s->N, s->J)) {
    const int SLICE = 16384;  // right-hand sides per launch (grid.y bound)
    for (int r0 = 0; r0 < nrhs; r0 += SLICE) {
      const int nr = std::min(SLICE, nrhs - r0);
      clr::SweepParams P;
      memset(&P, 0, sizeof(P));
      P.N = s->N; P.J = s->J; P.nrhs = nr;
      P.nchunk = wide ? clr::wdotl_chunks(s->N) : clr::sweep_chunks(s->N);
      P.L = (s->N - 1 + P.nchunk - 1) / P.nchunk;
      P.nchunk = (s->N - 1 + P.L - 1) / P.L;
      P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
      P.in = s->scratch.p + (size_t)r0 * s->N; P.out = s->scratch2.p + (size_t)r0 * s->N;
      if ((st = s->ws_elems.reserve((size_t)nr * P.nchunk * 3 * s->J)) != CLR_OK) return st;
      if (wide) clr::launch_wdotl_scan(P, s->ws_elems.p, s->stream);
      else clr::launch_dot_L_scan(P, s->ws_elems.p, s->stream);
    }
  } else {
    clr::launch_dot_L(s->N, s->J, nrhs, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                      s->scratch2.p, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(y, s->scratch2.p, sizeof(double) * (size_t)s->N * nrhs,
                         hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return CLR_OK;
}

int clr_solver_dot(clr_solver* s, double jitter, int n_a_real, const double* a_real,
                   int n_c_real, const double* c_real, int n_a_comp, const double* a_comp,
                   int n_b_comp, const double* b_comp, int n_c_comp, const double* c_comp,
                   int n_d_comp, const double* d_comp, int n_A, const double* A, int U_rows,
                   int U_cols, const double* U, int V_rows, int V_cols, const double* V, int n_x,
                   const double* x, int z_rows, int nrhs, const double* z, double* y) {
  const int N = z_rows;
  if (n_x != z_rows) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");  // :459
  int st = check_coeff_dims(n_a_real, n_c_real, n_a_comp, n_b_comp, n_c_comp, n_d_comp, n_A,
                            U_rows, U_cols, V_rows, V_cols, N);
  if (st != CLR_OK) return fail(st, "dimension mismatch");
  const bool has_general = (n_A != 0);
  const int J_general = U_rows, J_real = n_a_real, J_comp = n_a_comp;
  const int J = J_real + 2 * J_comp + J_general;
  if (J > CLR_MAX_WIDTH) return fail(CLR_UNSUPPORTED, "width above CLR_MAX_WIDTH");
  if (J_general > 0 && !has_general) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  if (N < 1 || nrhs < 1) return CLR_OK;

  if ((st = ensure_stream(s)) != CLR_OK) return st;
  hipStream_t stream = s->stream;
  const size_t total = (size_t)N * nrhs;

  if (J == 0) {  // cholesky.h:477-481: y = jitter * z (scaling done by the device copy engine
                 // would need a kernel; reuse the sweep with an all-zero width instead)
    if ((st = upload(s->scratch, z, total, stream)) != CLR_OK) return st;
    std::vector<double> dg((size_t)N, jitter);
    DevBuf tmp;
    if ((st = upload(tmp, dg.data(), (size_t)N, stream)) != CLR_OK) return st;
    if ((st = s->scratch2.reserve(total)) != CLR_OK) { tmp.release(); return st; }
    clr::launch_dot(N, 0, nrhs, nullptr, nullptr, nullptr, tmp.p, s->scratch.p, s->scratch2.p,
                    stream);
    hipError_t e = hipMemcpyAsync(y, s->scratch2.p, sizeof(double) * total,
                                  hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    tmp.release();
    if (e != hipSuccess) return fail(CLR_HIP_ERROR, hipGetErrorString(e));
    return CLR_OK;
  }

  // this call must not disturb a previously computed factor: use private buffers
  DevBuf coeffs, tt, dU, dV, phi, u, v, dg, zin, yout, ws;
  auto cleanup = [&]() {
    for (DevBuf* b : {&coeffs, &tt, &dU, &dV, &phi, &u, &v, &dg, &zin, &yout, &ws}) b->release();
  };
  std::vector<double> hc;
  double sum_ar = 0.0, sum_ac = 0.0;
  for (int j = 0; j < J_real; ++j) sum_ar += a_real[j];
  for (int j = 0; j < J_comp; ++j) sum_ac += a_comp[j];
  std::vector<double> hdg((size_t)N);
  for (int n = 0; n < N; ++n) {
    hdg[n] = (sum_ar + sum_ac) + jitter;  // cholesky.h:483-485
    if (has_general) hdg[n] += A[n];
  }
#define DOT_TRY(e)                \
  if ((st = (e)) != CLR_OK) {     \
    (void)hipStreamSynchronize(stream); \
    cleanup();                    \
    return st;                    \
  }
  DOT_TRY(upload_coeffs(coeffs, J_real, a_real, c_real, J_comp, a_comp, b_comp, c_comp, d_comp,
                        stream, hc));
  DOT_TRY(upload(tt, x, (size_t)N, stream));
  DOT_TRY(upload(dg, hdg.data(), (size_t)N, stream));
  DOT_TRY(upload(zin, z, total, stream));
  if (J_general) {
    DOT_TRY(upload(dU, U, (size_t)J_general * N, stream));
    DOT_TRY(upload(dV, V, (size_t)J_general * N, stream));
  }
  DOT_TRY(phi.reserve((size_t)J * N));
  DOT_TRY(u.reserve((size_t)J * N));
  DOT_TRY(v.reserve((size_t)J * N));
  DOT_TRY(yout.reserve(total));
  clr::GenericProblem g;
  g.N = N; g.J = J; g.J_real = J_real; g.J_comp = J_comp; g.J_general = J_general;
  g.a_real = coeffs.p; g.c_real = coeffs.p + J_real; g.a_comp = coeffs.p + 2 * J_real;
  g.b_comp = g.a_comp + J_comp; g.c_comp = g.b_comp + J_comp; g.d_comp = g.c_comp + J_comp;
  g.U = dU.p; g.V = dV.p; g.t = tt.p;
  clr::launch_dot_setup(g, phi.p, u.p, v.p, stream);
  if (clr::wdotl_scan_supported(N, J)) {  // long series: both triangles as chunked diagonal scans
    const int SLICE = 16384;
    for (int r0 = 0; r0 < nrhs; r0 += SLICE) {
      const int nr = std::min(SLICE, nrhs - r0);
      clr::SweepParams SP;
      memset(&SP, 0, sizeof(SP));
      SP.N = N; SP.J = J; SP.nrhs = nr;
      SP.nchunk = clr::wdotl_chunks(N);
      SP.L = (N - 1 + SP.nchunk - 1) / SP.nchunk;
      SP.nchunk = (N - 1 + SP.L - 1) / SP.L;
      SP.phi = phi.p; SP.u = u.p;
      SP.in = zin.p + (size_t)r0 * N; SP.out = yout.p + (size_t)r0 * N;
      DOT_TRY(ws.reserve((size_t)nr * SP.nchunk * 3 * J));
      clr::launch_wdot_scan(SP, v.p, dg.p, ws.p, stream);
    }
  } else {
    clr::launch_dot(N, J, nrhs, phi.p, u.p, v.p, dg.p, zin.p, yout.p, stream);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(y, yout.p, sizeof(double) * total, hipMemcpyDeviceToHost, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  cleanup();
#undef DOT_TRY
  if (e != hipSuccess) return fail(CLR_HIP_ERROR, hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(CLR_HIP_ERROR, hipGetErrorString(e2));
  return CLR_OK;
}

int clr_solver_predict(const clr_solver* cs, int n_y, const double* y, int M, const double* xs,
                       double* pred) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  int st = sweep_common(s, n_y, 1, y);  // also checks N / computed (:600-601)
  if (st != CLR_OK) return st;
  if (M <= 0) return CLR_OK;
  if (s->coeffs_lazy) {  // (the one-launch compute passed the coefficients as kernel arguments)
    if ((st = upload(s->coeffs, s->host_coeffs.data(), s->host_coeffs.size(), s->stream)) != CLR_OK) return st;
    s->coeffs_lazy = false;
  }
  if (s->t.cap < (size_t)s->N || s->coeffs.p == nullptr)
    return fail(CLR_UNSUPPORTED,
                "predict needs the inputs of compute(); a solver restored from a pickled "
                "state does not carry them (same as the reference, solver.cpp:36-42)");
  hipStream_t stream = s->stream;
  // alpha = K^-1 y  (:608)
  if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, 1, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;
    if ((st = sweep_scan(s, 1, s->scratch2.p, s->scratch2.p, nullptr, 1)) != CLR_OK) return st;
  } else {
    clr::launch_solve(s->N, s->J, 1, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                      s->scratch2.p, stream);
  }
  // the reference's two passes walk the prediction points in order (:616-653,657-695):
  // with sorted points both passes become scans + one thread per point
  bool sorted = true;
  for (int m = 1; m < M && sorted; ++m) sorted = xs[m - 1] <= xs[m];
  const bool scan = sorted && clr::predict_scan_supported(s->N, s->J_real, s->J_comp);
  int pchunk = 0, pL = 0;
  if (scan) {
    pchunk = clr::sweep_chunks(s->N);
    pL = (s->N + pchunk - 1) / pchunk;
    pchunk = (s->N + pL - 1) / pL;
    if ((st = s->ws_elems.reserve(clr::predict_workspace_doubles(pchunk, s->J_real + 2 * s->J_comp))) != CLR_OK) return st;
  }
  DevBuf dxs, dpred;
  if ((st = upload(dxs, xs, (size_t)M, stream)) != CLR_OK) return st;
  if ((st = dpred.reserve((size_t)M)) != CLR_OK) { dxs.release(); return st; }
  hipError_t e = hipMemsetAsync(dpred.p, 0, sizeof(double) * (size_t)M, stream);
  const clr::GenericProblem g = generic_view(s);
  if (scan) clr::launch_predict_scan(g, s->scratch2.p, M, dxs.p, dpred.p, s->ws_elems.p, pchunk, pL, stream);
  else clr::launch_predict(g, s->scratch2.p, M, dxs.p, dpred.p, stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(pred, dpred.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  dxs.release();
  dpred.release();
  if (e != hipSuccess) return fail(CLR_HIP_ERROR, hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(CLR_HIP_ERROR, hipGetErrorString(e2));
  return CLR_OK;
}

int clr_solver_get_dims(const clr_solver* s, int* computed, int* N, int* J, double* log_det) {
  if (computed) *computed = s->computed;
  if (N) *N = s->N;
  if (J) *J = s->J;
  if (log_det) *log_det = s->log_det;
  return CLR_OK;
}

int clr_solver_get_state(const clr_solver* cs, double* phi, double* u, double* W, double* D) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  if (!s->computed) return fail(CLR_NOT_COMPUTED, "you must call 'compute' first");
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  const size_t N = (size_t)s->N, J = (size_t)s->J, Nm1 = N - 1;
  if (J * Nm1) {
    HIP_TRY(hipMemcpyAsync(phi, s->phi.p, sizeof(double) * J * Nm1, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(u, s->u.p, sizeof(double) * J * Nm1, hipMemcpyDeviceToHost, s->stream));
  }
  if (J * N)
    HIP_TRY(hipMemcpyAsync(W, s->W.p, sizeof(double) * J * N, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(D, s->D.p, sizeof(double) * N, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return CLR_OK;
}

int clr_solver_set_state(clr_solver* s, int computed, int N, int J, double log_det,
                         const double* phi, const double* u, const double* W, const double* D) {
  // solver.cpp:44-58: plain member assignment; coefficients and t are NOT part
  // of the state (so predict is unavailable afterwards, as in the reference).
  s->computed = 0;
  s->N = N;
  s->J = J;
  s->log_det = log_det;
  s->J_real = s->J_comp = s->J_general = 0;
  if (!computed) return CLR_OK;
  if (J < 0 || J > CLR_MAX_WIDTH || N < 1) return fail(CLR_INVALID_ARGUMENT, "Invalid state!");
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  const size_t Nn = (size_t)N, Jn = (size_t)J, Nm1 = Nn - 1;
  if ((st = upload(s->phi, phi, Jn * Nm1, s->stream)) != CLR_OK) return st;
  if ((st = upload(s->u, u, Jn * Nm1, s->stream)) != CLR_OK) return st;
  if ((st = upload(s->W, W, Jn * Nn, s->stream)) != CLR_OK) return st;
  if ((st = upload(s->D, D, Nn, s->stream)) != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->coeffs.release();  // marks "inputs unknown" for predict
  s->computed = 1;
  return CLR_OK;
}

/* ---- batched log-likelihood ---------------------------------------------------- */
clr_batch* clr_batch_create(int B, int N, int J_real, int J_comp, int device) {
  if (B < 1 || N < 1 || J_real < 0 || J_comp < 0) {
    fail(CLR_INVALID_ARGUMENT, "clr_batch_create: bad sizes");
    return nullptr;
  }
  const clr::BatchLaunchers* L = clr::find_batch_launchers(J_real, J_comp);
  const int width = J_real + 2 * J_comp;
  if (!L && (width < 1 || width > clr::wide_max_width())) {
    fail(CLR_UNSUPPORTED, "batched path supports widths 1..64 (J_real + 2 J_comp)");
    return nullptr;
  }
  if (require_device(device) != CLR_OK) return nullptr;
  clr_batch* h = new clr_batch();
  h->device = device;
  h->B = B;
  h->N = N;
  h->J_real = J_real;
  h->J_comp = J_comp;
  h->J = J_real + 2 * J_comp;
  h->launch = L;  // null: widths 9..64, one wave per problem (wide_kernels.hip)
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    fail(CLR_HIP_ERROR, "hipStreamCreate failed");
    delete h;
    return nullptr;
  }
  if (clr_batch_set_chunks(h, 0) != CLR_OK) {
    clr_batch_destroy(h);
    return nullptr;
  }
  return h;
}

void clr_batch_destroy(clr_batch* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (DevBuf* b : {&h->coeffs, &h->t, &h->diag, &h->y, &h->tT, &h->dT, &h->yT,
                    &h->elems, &h->starts, &h->part, &h->partx, &h->cond, &h->out, &h->phi, &h->u, &h->W, &h->D,
                    &h->fphi, &h->fu, &h->fW, &h->fD, &h->lvl_elems, &h->lvl_starts, &h->wstarts, &h->wends,
                    &h->wpart, &h->wresid, &h->wT, &h->wD, &h->wY, &h->gA, &h->gU, &h->gV, &h->g_riders, &h->g_out,
                    &h->g_res, &h->g_rec, &h->g_ck})
    b->release();
  if (h->flags) (void)hipFree(h->flags);
  if (h->wints) (void)hipFree(h->wints);
  if (h->g_ckflag) (void)hipFree(h->g_ckflag);
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  if (h->pin) (void)hipHostFree(h->pin);
  clr::staging_destroy(h->staging);
  h->scan.release();
  for (DevBuf* b : {&h->gen_elems, &h->gen_starts, &h->gen_part, &h->gen_cond}) b->release();
  if (h->gen_flags) (void)hipFree(h->gen_flags);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int warm_plan_chunks(clr_batch* h);
static int warm_resolve(clr_batch* h, bool* pin_current);
static int warm_scan_spans(clr_batch* h);
static void warm_select(clr_batch* h);

int clr_batch_set_chunks(clr_batch* h, int nchunk) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  h->warm_explicit_chunks = nchunk > 0 ? nchunk : 0;
  if (nchunk > 0) h->pipeline_pinned = true;
  if (!h->launch) {
    // wide path: one wave per (problem, chunk).  One chunk (the plain sequential sweep)
    // unless the batch alone leaves the chip underused: then ~2 waves per SIMD worth of
    // chunks, at ~1.25x the work per sample (profiles/r01s); widths above 32 stay sequential
    if (h->J > clr::wide_scan_max_width()) nchunk = 1;
    else if (nchunk <= 0) {
      // One round of two waves per SIMD: B x nchunk = 2048 waves.  (Round 2 used 4096 / B -- two rounds of half the
      //  length -- because the checked replay of borderline problems, 7 ms for config 4, got shorter with the chunks;
      //  with the round-3 routing that family is settled from the chunk summaries and the sequential prefix + the
      //  correct phase decide: B = 256: 8 chunks 17.1 ms, 16 chunks 18.5, 10 chunks 22.8 (2560 waves = a second,
      //  nearly empty round), B = 512: 4 chunks 32.2 ms, 8 chunks 32.7; profiles/r03_wide_chunks.txt)
      nchunk = h->B <= 1024 ? 2048 / h->B : 1;  // (above 1024 problems one sweep per problem already fills a round)
      if (nchunk < 2) nchunk = 1;
      if (nchunk > 16) nchunk = 16;
      while (nchunk > 1 && h->N / nchunk < 512) --nchunk;
    }
    if (nchunk > h->N / 64) nchunk = std::max(1, h->N / 64);
  } else if (nchunk <= 0) {
    nchunk = auto_chunks(h->B, h->N, h->J);
  }
  if (nchunk > h->N) nchunk = h->N;
  h->L = (h->N + nchunk - 1) / nchunk;
  if (nchunk > 1 && h->L > 8) h->L = (h->L + 7) & ~7;  // 64-B aligned chunk rows for the tile loads
  h->nchunk = (h->N + h->L - 1) / h->L;
  h->L0 = 0;
  if (const char* e = getenv("CLR_WIDE_FIRST_RATIO")) h->wide_first_ratio = atof(e);  // (tuning runs only)
  if (!h->launch && h->nchunk > 1 && h->wide_first_ratio > 1.0) {
    // wide scan: the first chunk's summarize carries no riders (wide_scan_body, RIDERS == false) and costs
    // ~1 / wide_first_ratio of a later chunk's per sample: it gets that many more samples, so that all waves of the
    // one round finish together.  Chunks 1.. have exactly L samples, the first one the rest.
    const int nc = h->nchunk;
    int L = (int)ceil(h->N / (nc - 1 + h->wide_first_ratio));
    L = (L + 7) & ~7;
    const long first = (long)h->N - (long)(nc - 1) * L;
    if (L >= 64 && first >= L) { h->L = L; h->L0 = (int)first; }
  }
  h->relayout_pending = true;
  h->grad_span_valid = false;
  h->have_factor = false;  // its layout depends on the chunking
  const size_t pc = (size_t)h->B * h->nchunk;
  h->plan = clr::plan_prefix(h->nchunk, 0, 0);
  if (h->launch) {
    if ((st = h->elems.reserve(pc * h->launch->elem_doubles)) != CLR_OK) return st;
    if ((st = h->starts.reserve(pc * h->launch->start_doubles)) != CLR_OK) return st;
    h->plan = clr::plan_prefix(h->nchunk, h->plan_levels, h->plan_g, h->B, h->J);
    size_t le = 0, ls = 0;
    clr::multilevel_workspace(h->plan, h->J, &le, &ls);
    if (le && (st = h->lvl_elems.reserve((size_t)h->B * le)) != CLR_OK) return st;
    if (ls && (st = h->lvl_starts.reserve((size_t)h->B * ls)) != CLR_OK) return st;
  } else if (h->nchunk > 1) {  // elements / start states at the padded width (16 or 32)
    const size_t JP = h->J <= 16 ? 16 : 32, SZ = JP * (JP + 1) / 2;
    if ((st = h->elems.reserve(pc * (JP * JP + JP + SZ + JP + SZ))) != CLR_OK) return st;
    if ((st = h->starts.reserve(pc * (SZ + JP))) != CLR_OK) return st;
  }
  if ((st = h->part.reserve(pc * 2)) != CLR_OK) return st;
  if ((st = h->partx.reserve(pc * 2)) != CLR_OK) return st;
  if ((st = h->cond.reserve(pc * 4)) != CLR_OK) return st;  // gamma, mu, residual per chunk | measured G error
  if ((st = h->out.reserve((size_t)h->B * 3 + ((size_t)h->B + 1) / 2)) != CLR_OK) return st;
  if (h->flags) (void)hipFree(h->flags);
  h->flags = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->flags), (2 * pc + (size_t)h->B) * sizeof(int)));
  // a single-chunk plan launches no prefix / correct kernel: nothing else would ever clear need_exact or fill
  // the conditioning record
  HIP_TRY(hipMemsetAsync(h->flags, 0, (2 * pc + (size_t)h->B) * sizeof(int), h->stream));
  HIP_TRY(hipMemsetAsync(h->cond.p, 0, pc * 4 * sizeof(double), h->stream));
  h->evaluated = false;
  if ((st = warm_plan_chunks(h)) != CLR_OK) return st;
  return CLR_OK;
}

// The warm path's chunking: two waves per SIMD worth of (problem, chunk) lanes (the plain recurrence needs 189
// registers), chunks of at least 256 samples (the warm-up is at most half a chunk); an explicit chunk count is
// honoured (results then do not depend on the batch size, i.e. on a sharding).
static int warm_plan_chunks(clr_batch* h) {
  h->wnchunk = 0;
  h->wL = 0;
  h->warm_active = false;
  h->warm_span.clear();  // (the spans belong to a chunking: rescanned by the next set_series)
  if (!h->launch || h->N < 512) return CLR_OK;
  long want = h->warm_explicit_chunks ? h->warm_explicit_chunks : std::max<long>(1, 131072 / h->B);
  long L = (h->N + want - 1) / want;
  if (L < 256) L = 256;
  L = (L + 7) & ~7L;
  const long nc = (h->N + L - 1) / L;
  if (nc < 2) return CLR_OK;
  h->wL = (int)L;
  h->wnchunk = (int)nc;
  h->wKpad = std::min(128, h->wL / 2);  // rows of warm-up every chunk's column carries (the largest candidate)
  h->wrows = h->wKpad + h->wL + 8;
  h->warm_copy_pending = true;
  const size_t pc = (size_t)h->B * nc, START = (size_t)h->launch->start_doubles;
  int st;
  if ((st = h->wstarts.reserve(pc * START)) != CLR_OK) return st;
  if ((st = h->wends.reserve(pc * START)) != CLR_OK) return st;
  if ((st = h->wpart.reserve(pc * 2)) != CLR_OK) return st;
  if ((st = h->wresid.reserve((size_t)h->B)) != CLR_OK) return st;
  const size_t ints = pc + 2 * (size_t)h->B;
  if (ints > h->wints_cap) {
    if (h->wints) (void)hipFree(h->wints);
    h->wints = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->wints), ints * sizeof(int)));
    h->wints_cap = ints;
  }
  HIP_TRY(hipMemsetAsync(h->wints, 0, ints * sizeof(int), h->stream));
  if (h->have_series) {  // the spans and the warm-ups follow the new chunking (the series are resident)
    if ((st = warm_scan_spans(h)) != CLR_OK) return st;
    warm_select(h);
  }
  return CLR_OK;
}

// For every problem (or the one shared series) and every candidate K: the shortest time the K samples in front of
// a chunk boundary of the warm path span.  O(B x chunks) lookups into the series resident in HBM (warm_spans_kernel),
// so the spans follow the chunking too (clr_batch_set_chunks after clr_batch_set_series).
static int warm_scan_spans(clr_batch* h) {
  h->warm_span.clear();
  if (h->wnchunk < 2 || !h->have_series) return CLR_OK;
  const int nb = h->t_stride == 0 ? 1 : h->B;
  const size_t n = (size_t)nb * clr_batch::WARM_NK;
  int st;
  if ((st = h->scan.reserve(std::max(n, (size_t)nb * 4))) != CLR_OK) return st;
  clr::WarmCands cands;
  cands.nk = clr_batch::WARM_NK;
  for (int k = 0; k < clr_batch::WARM_NK; ++k) cands.K[k] = h->warm_cand[k];
  clr::launch_warm_spans(h->t.p, h->t_stride, nb, h->wL, h->wnchunk, cands, h->scan.p, h->stream);
  h->warm_span.assign(n, 0.0);
  HIP_TRY(hipMemcpyAsync(h->warm_span.data(), h->scan.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CLR_OK;
}

int clr_batch_get_chunks(const clr_batch* h, int* nchunk, int* chunk_len) {
  if (nchunk) *nchunk = h->nchunk;
  if (chunk_len) *chunk_len = h->L;
  return CLR_OK;
}

// Warm-started recurrence: warm-up steps per problem from its slowest decay rate (host_cmin, kept from the last
// set_coefficients) and the time the samples in front of its chunk boundaries span (warm_span, from the last
// set_series): exp(-c_min x span) <= exp(-32) = 1.3e-14 -- what is left of ANY start state after the warm-up by the
// decay alone, three orders below the tolerance of the boundary check (the update by the data only forgets faster); the
// check of warm_check_kernel certifies the choice.  Called whenever either side changes, so that the K in force always
// belongs to the (series, coefficients) pair in force.
static void warm_select(clr_batch* h) {
  const size_t B = (size_t)h->B;
  h->warm_active = false;
  h->warm_K_dirty = false;
  if (h->warm_mode == 0 || h->wnchunk < 2 || !h->have_series || h->warm_span.empty() || h->host_cmin.size() != B) return;
  h->warm_K.assign(B, 0);
  size_t eligible = 0;
  const bool shared = h->t_stride == 0;
  for (size_t b = 0; b < B; ++b) {
    int K = 0;
    if (h->warm_mode == 1) {
      K = std::min(h->warm_forced_K, h->wL / 2);
    } else {
      const double cmin = h->host_cmin[b];
      const double* span = &h->warm_span[(shared ? 0 : b) * clr_batch::WARM_NK];
      for (int k = 0; k < clr_batch::WARM_NK && cmin > 0.0; ++k)
        if (cmin * span[k] >= 32.0) {
          K = h->warm_cand[std::min(k + h->warm_boost, clr_batch::WARM_NK - 1)];
          if (K > h->wL / 2) K = 0;
          break;
        }
    }
    h->warm_K[b] = K;
    eligible += K > 0;
  }
  // (a batch with only a few eligible problems is not worth a second set of launches)
  h->warm_active = eligible * 2 >= B;
  h->warm_K_dirty = h->warm_active;  // (uploaded behind the next coefficients, or by the next enqueue)
}

int clr_batch_set_series(clr_batch* h, const double* t, long t_stride, const double* diag,
                         long diag_stride, const double* y, long y_stride) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  const long N = h->N;
  for (long sd : {t_stride, diag_stride, y_stride})
    if (sd != 0 && sd != N)
      return fail(CLR_INVALID_ARGUMENT, "series stride must be 0 (shared) or N");
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;  // (an evaluation in flight is settled on ITS series)
  auto count = [&](long sd) { return (size_t)(sd == 0 ? N : N * (long)h->B); };
  const auto host_t0 = std::chrono::steady_clock::now();
  if ((st = h->t.reserve(count(t_stride))) != CLR_OK) return st;
  if ((st = h->diag.reserve(count(diag_stride))) != CLR_OK) return st;
  if ((st = h->y.reserve(count(y_stride))) != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(h->stream));  // (kernels of an earlier evaluation may still be reading the old series)
  const clr::CopyJob jobs[3] = {{h->t.p, t, count(t_stride)}, {h->diag.p, diag, count(diag_stride)}, {h->y.p, y, count(y_stride)}};
  const size_t total = (jobs[0].n + jobs[1].n + jobs[2].n) * sizeof(double);
  if (total >= ((size_t)32 << 20)) {
    // large series: NT host threads stage pieces through pinned buffers, their DMAs share the link (clr_series_io.h)
    int e = clr::staging_create(h->staging, h->device);
    if (e == 0) e = clr::upload_parallel(h->staging, jobs, 3);
    if (e != 0) return fail(CLR_HIP_ERROR, hipGetErrorString((hipError_t)e));
  } else {
    for (const clr::CopyJob& j : jobs)
      if (j.n) HIP_TRY(hipMemcpyAsync(j.dst, j.src, j.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  h->t_stride = t_stride;
  h->diag_stride = diag_stride;
  h->y_stride = y_stride;
  h->have_series = true;
  // one pass over t ON THE DEVICE: max |t| over every sample (sortedness is not assumed), the largest and the smallest
  // step, NaN times; then the warm path's spans
  {
    const int nb = t_stride == 0 ? 1 : h->B;
    if ((st = h->scan.reserve((size_t)nb * std::max(4, (int)clr_batch::WARM_NK))) != CLR_OK) return st;
    clr::launch_series_stats(h->t.p, t_stride, nb, (int)N, h->scan.p, h->stream);
    std::vector<double> stats((size_t)nb * 4);
    HIP_TRY(hipMemcpyAsync(stats.data(), h->scan.p, stats.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double tm = 0.0, dm = 0.0, dmin = INFINITY;
    bool nan = false;
    for (int b = 0; b < nb; ++b) {
      tm = std::max(tm, stats[4 * b]); dm = std::max(dm, stats[4 * b + 1]); dmin = std::min(dmin, stats[4 * b + 2]);
      nan = nan || stats[4 * b + 3] != 0.0;
    }
    // (a NaN time: NaN bounds select the conservative kernels, sel_max)
    h->tmax = nan ? NAN : tm;
    h->dxmax = nan ? NAN : dm;
    h->dtmin = nan ? NAN : (N > 1 ? dmin : 0.0);
  }
  if ((st = warm_scan_spans(h)) != CLR_OK) return st;
  h->set_series_host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  h->grad_span_valid = false;
  h->relayout_pending = true;
  h->warm_copy_pending = true;
  // a new series: the warm-ups chosen for the previous one's spans do not apply, nor does its history of fallbacks
  if (h->warm_mode < 0) h->warm_boost = 0;
  warm_select(h);
  return CLR_OK;
}

int clr_batch_get_series_order(const clr_batch* h, double* dtmin) {
  if (!h->have_series) return fail(CLR_INVALID_ARGUMENT, "no series set");
  if (dtmin) *dtmin = h->dtmin;
  return CLR_OK;
}

int clr_batch_clear_series(clr_batch* h) {
  int st = warm_resolve(h, nullptr);
  if (st != CLR_OK) return st;
  h->have_series = false;
  h->warm_active = false;
  h->warm_span.clear();
  return CLR_OK;
}

int clr_batch_get_selection_bounds(const clr_batch* h, double* tmax, double* dxmax, double* dmax, double* cmax,
                                   double* set_series_host_ms) {
  if (tmax) *tmax = h->tmax;
  if (dxmax) *dxmax = h->dxmax;
  if (dmax) *dmax = h->dmax;
  if (cmax) *cmax = h->cmax;
  if (set_series_host_ms) *set_series_host_ms = h->set_series_host_ms;
  return CLR_OK;
}

int clr_batch_set_selection_bounds(clr_batch* h, double tmax, double dxmax, double dmax, double cmax) {
  // negative: leave that floor as it is (NaN counts as "unbounded": the conservative kernels)
  if (!(tmax < 0.0)) h->floor_tmax = tmax;
  if (!(dxmax < 0.0)) h->floor_dxmax = dxmax;
  if (!(dmax < 0.0)) h->floor_dmax = dmax;
  if (!(cmax < 0.0)) h->floor_cmax = cmax;
  return CLR_OK;
}

static int reserve_pinned(clr_batch* h, size_t doubles) {
  if (doubles <= h->pin_cap && h->pin) return CLR_OK;
  if (h->pin) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_cap = 0;
  }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin), doubles * sizeof(double), hipHostMallocDefault));
  h->pin_cap = doubles;
  return CLR_OK;
}

int clr_batch_set_coefficients(clr_batch* h, const double* jitter, const double* a_real,
                               const double* c_real, const double* a_comp, const double* b_comp,
                               const double* c_comp, const double* d_comp) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;  // (pending problems of the evaluation in flight: at ITS coefficients)
  const size_t B = (size_t)h->B, nr = B * h->J_real, nc = B * h->J_comp;
  h->dmax = 0.0;
  h->cmax = 0.0;
  for (size_t i = 0; i < nc; ++i) {
    const double m = fabs(d_comp[i]), c = fabs(c_comp[i]);
    if (!(m <= h->dmax)) h->dmax = m;
    if (!(c <= h->cmax)) h->cmax = c;
  }
  for (size_t i = 0; i < nr; ++i) {
    const double c = fabs(c_real[i]);
    if (!(c <= h->cmax)) h->cmax = c;
  }
  h->host_cmax.assign(B, 0.0);
  for (size_t b = 0; b < B; ++b) {
    double m = 0.0;
    for (int j = 0; j < h->J_real; ++j) m = std::max(m, fabs(c_real[b * h->J_real + j]));
    for (int j = 0; j < h->J_comp; ++j) m = std::max(m, fabs(c_comp[b * h->J_comp + j]));
    h->host_cmax[b] = m;
  }
  h->host_cmin.assign(B, INFINITY);
  for (size_t b = 0; b < B; ++b) {
    double cmin = INFINITY;
    for (int j = 0; j < h->J_real; ++j) { const double c = c_real[b * h->J_real + j]; if (!(c >= cmin)) cmin = c; }
    for (int j = 0; j < h->J_comp; ++j) { const double c = c_comp[b * h->J_comp + j]; if (!(c >= cmin)) cmin = c; }
    h->host_cmin[b] = cmin;
  }
  warm_select(h);
  // one pinned staging buffer, one copy: a_real c_real a_comp b_comp c_comp d_comp | jitter
  const size_t total = 2 * nr + 4 * nc + B;
  if ((st = reserve_pinned(h, std::max(total, 3 * B + (B + 1) / 2) + (B + 1) / 2)) != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(h->stream));  // (a previous upload may still read the staging buffer)
  double* w = h->pin;
  auto put = [&](const double* p, size_t n) { if (n) memcpy(w, p, n * sizeof(double)); w += n; };
  put(a_real, nr); put(c_real, nr); put(a_comp, nc); put(b_comp, nc); put(c_comp, nc); put(d_comp, nc);
  if (jitter) { put(jitter, B); h->host_jitter.assign(jitter, jitter + B); }
  else { memset(w, 0, B * sizeof(double)); w += B; h->host_jitter.assign(B, 0.0); }  // NULL: no jitter
  if ((st = h->coeffs.reserve(total)) != CLR_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->coeffs.p, h->pin, total * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->warm_active) {  // K per problem, behind the coefficients in the staging buffer
    int* kk = reinterpret_cast<int*>(h->pin + std::max(total, 3 * B + (B + 1) / 2));
    memcpy(kk, h->warm_K.data(), B * sizeof(int));
    HIP_TRY(hipMemcpyAsync(h->wints + (size_t)h->B * h->wnchunk + B, kk, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    h->warm_K_dirty = false;
  }
  h->have_coeffs = true;
  return CLR_OK;
}

// the maxima the kernel selection looks at: the plan's own, raised to the floors of a sharded parent
static double sel_max(double own, double floor) {  // (NaN on either side wins: the conservative kernels)
  if (own != own) return own;
  if (floor != floor) return floor;
  return own >= floor ? own : floor;
}
static bool lazy_eligible(const clr_batch* h) {
  // |c dx| < 2^-7 at every step: Psi stays within [0.88, 1] over the 16 steps between renormalisations;
  // |d dx| < 2^-5: the per-step rotation of the (cos, sin) pairs uses a short Taylor series
  const double cmax = sel_max(h->cmax, h->floor_cmax), dmax = sel_max(h->dmax, h->floor_dmax),
               dxmax = sel_max(h->dxmax, h->floor_dxmax);
  return h->have_series && h->have_coeffs && cmax * dxmax < 0.0078125 && dmax * dxmax < 0.03125;
}

static bool split_active(const clr_batch* h) {
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

static int batch_params(clr_batch* h, int materialize, clr::BatchParams& P) {
  if (!h->have_series || !h->have_coeffs)
    return fail(CLR_INVALID_ARGUMENT, "set_series and set_coefficients must be called first");
  int st = CLR_OK;
  if (materialize && !h->have_factor) {
    // widths 1..8: chunk-interleaved device layout (replay_chunk, MATERIALIZE == 2); widths 9..64: the wide kernels
    // write the reference's own storage per problem (wide_scan_kernel, MODE 0: phi, u [N-1][J], W [N][J], D [N])
    const size_t B = (size_t)h->B, J = (size_t)h->J, cells = h->launch ? (size_t)h->L * h->nchunk : (size_t)h->N;
    if ((st = h->phi.reserve(B * J * cells)) != CLR_OK) return st;
    if ((st = h->u.reserve(B * J * cells)) != CLR_OK) return st;
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
  if (!h->launch && h->nchunk > 1 && (h->summarize_mode < 0 || h->summarize_mode == 2) && lazy_eligible(h))
    P.split_lazy = 1;
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
  P.wide_materialize = (materialize && !h->launch) ? 1 : 0;
  P.phi = h->phi.p; P.u = h->u.p; P.W = h->W.p; P.D = h->D.p;
  return CLR_OK;
}

// The replay's view of the series.  The role-split summarize reads the chunk-interleaved copy; the replay is free to
// read either that copy (one 512-B line per array and step per wave, but a second 2.4 GB stream competing with the
// factor's stores) or the row-major arrays through the LDS-staged tiles (round 1's path).
static clr::BatchParams replay_view(const clr_batch* h, const clr::BatchParams& P, int materialize) {
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
static bool batch_relayout(clr_batch* h) {
  if (!((h->layout == 1 || split_active(h)) && h->nchunk > 1)) return false;
  const long cells = (long)h->nchunk * h->L;
  struct { DevBuf* src; DevBuf* dst; long stride; int pad; } jobs[3] = {
      {&h->t, &h->tT, h->t_stride, 1}, {&h->diag, &h->dT, h->diag_stride, 2}, {&h->y, &h->yT, h->y_stride, 0}};
  for (auto& j : jobs)
    clr::launch_relayout(j.src->p, j.stride, j.dst->p, j.stride ? cells : 0, j.stride ? h->B : 1,
                         h->N, h->L, h->nchunk, j.pad, h->stream);
  return true;
}

int clr_batch_set_exact(clr_batch* h, int force) {
  h->force_exact = force ? 1 : 0;
  return CLR_OK;
}

int clr_batch_get_exact_count(clr_batch* h, int* count) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!count) return fail(CLR_INVALID_ARGUMENT, "count is null");
  if (h->nchunk < 2 || h->force_exact) {  // every problem went through the reference recurrence
    *count = h->B;
    return CLR_OK;
  }
  std::vector<int> need((size_t)h->B);
  const size_t pc = (size_t)h->B * h->nchunk;
  HIP_TRY(hipMemcpyAsync(need.data(), h->flags + 2 * pc, need.size() * sizeof(int),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int n = 0;
  for (int v : need) n += v != 0;
  *count = n;
  return CLR_OK;
}

int clr_batch_get_exact_flags(clr_batch* h, int* flags) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!flags) return fail(CLR_INVALID_ARGUMENT, "flags is null");
  if (h->nchunk < 2) {
    for (int b = 0; b < h->B; ++b) flags[b] = 2;  // one chunk: the replay from the zero state is the recurrence
    return CLR_OK;
  }
  const size_t pc = (size_t)h->B * h->nchunk;
  HIP_TRY(hipMemcpyAsync(flags, h->flags + 2 * pc, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->force_exact)
    for (int b = 0; b < h->B; ++b) flags[b] = flags[b] < 1 ? 1 : flags[b];
  return CLR_OK;
}

int clr_batch_get_conditioning_chunkwise(clr_batch* h, double* ratio_max) {
  // max over chunks of gamma_c / mu_c (both of the SAME chunk), per problem
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!ratio_max) return fail(CLR_INVALID_ARGUMENT, "ratio_max is null");
  if (!h->evaluated) return fail(CLR_NOT_COMPUTED, "no evaluation has been enqueued on this plan");
  if (h->nchunk < 2) {  // one chunk: the recurrence itself ran, there is no record
    for (int b = 0; b < h->B; ++b) ratio_max[b] = 0.0;
    return CLR_OK;
  }
  const size_t pc = (size_t)h->B * h->nchunk;
  std::vector<double> c(pc * 3);
  HIP_TRY(hipMemcpyAsync(c.data(), h->cond.p, c.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; ++b) {
    double r = 0.0;
    for (int k = 0; k < h->nchunk; ++k) {
      const double* e = &c[((size_t)b * h->nchunk + k) * 3];
      const double q = e[0] / e[1];
      if (!(q <= r)) r = q;
    }
    ratio_max[b] = r;
  }
  return CLR_OK;
}

int clr_batch_get_conditioning(clr_batch* h, double* gamma_max, double* mu_min, double* resid_max) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->evaluated) return fail(CLR_NOT_COMPUTED, "no evaluation has been enqueued on this plan");
  if (h->nchunk < 2) {  // one chunk: the recurrence itself ran, there is no record
    for (int b = 0; b < h->B; ++b) {
      if (gamma_max) gamma_max[b] = 0.0;
      if (mu_min) mu_min[b] = 1.0;
      if (resid_max) resid_max[b] = 0.0;
    }
    return CLR_OK;
  }
  const size_t pc = (size_t)h->B * h->nchunk;
  std::vector<double> c(pc * 3);
  HIP_TRY(hipMemcpyAsync(c.data(), h->cond.p, c.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  // problems the warm-started recurrence settled have no scan record: gamma 0, mu 1, and the largest boundary
  // mismatch of the warm path as the residual
  std::vector<int> scanned;
  std::vector<double> wres;
  if (h->warm_active && h->wints && h->warm_settled > 0) {
    scanned.resize((size_t)h->B);
    wres.resize((size_t)h->B);
    HIP_TRY(hipMemcpyAsync(scanned.data(), h->wints + (size_t)h->B * h->wnchunk, scanned.size() * sizeof(int),
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(wres.data(), h->wresid.p, wres.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  for (int b = 0; b < h->B; ++b) {
    double g = 0.0, m = 1.0, r = 0.0;
    if (!scanned.empty() && scanned[b] == 0) {
      r = wres[b];
    } else {
      for (int k = 0; k < h->nchunk; ++k) {
        const double* e = &c[((size_t)b * h->nchunk + k) * 3];
        if (!(e[0] <= g)) g = e[0];
        if (!(e[1] >= m)) m = e[1];
        if (!(e[2] <= r)) r = e[2];
      }
    }
    if (gamma_max) gamma_max[b] = g;
    if (mu_min) mu_min[b] = m;
    if (resid_max) resid_max[b] = r;
  }
  return CLR_OK;
}

int clr_batch_get_measured_error(clr_batch* h, double* eg_max) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!eg_max) return fail(CLR_INVALID_ARGUMENT, "eg_max is null");
  if (!h->evaluated) return fail(CLR_NOT_COMPUTED, "no evaluation has been enqueued on this plan");
  const size_t pc = (size_t)h->B * h->nchunk;
  std::vector<double> c(pc);
  if (h->nchunk >= 2) {
    HIP_TRY(hipMemcpyAsync(c.data(), h->cond.p + pc * 3, pc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  for (int b = 0; b < h->B; ++b) {
    double e = 0.0;
    for (int k = 0; k < h->nchunk && h->nchunk >= 2; ++k) {
      const double v = c[(size_t)b * h->nchunk + k];
      if (!(v <= e)) e = v;
    }
    eg_max[b] = e;
  }
  return CLR_OK;
}

int clr_batch_set_certificate(clr_batch* h, double max_gamma_over_mu, double max_residual) {
  h->cert_gamma = max_gamma_over_mu;
  h->cert_resid = max_residual;
  h->pipeline_pinned = true;
  return CLR_OK;
}

int clr_batch_set_certificate_gamma(clr_batch* h, double max_gamma, double max_gamma_times_error) {
  h->cert_gamma_abs = max_gamma;
  h->cert_eg = max_gamma_times_error;
  return CLR_OK;
}

int clr_batch_set_prefix_mode(clr_batch* h, int mode) {
  if (mode < 0 || mode > 2) return fail(CLR_INVALID_ARGUMENT, "prefix mode must be 0, 1 or 2");
  h->coop_prefix = mode;
  h->pipeline_pinned = true;
  return CLR_OK;
}

int clr_batch_set_prefix_plan(clr_batch* h, int levels, int group) {
  if (levels > 3 || (levels > 0 && group < 2)) return fail(CLR_INVALID_ARGUMENT, "prefix plan: levels <= 3, group >= 2");
  h->plan_levels = levels;
  h->plan_g = group;
  const int keep = h->warm_explicit_chunks;  // (re-planning the workspace is not a request for a chunk count)
  int st = clr_batch_set_chunks(h, h->nchunk);
  h->warm_explicit_chunks = keep;
  if (st == CLR_OK) st = warm_plan_chunks(h);
  return st;
}

int clr_batch_get_prefix_plan(const clr_batch* h, int* levels, int* groups /* [3] */, int* counts /* [4] */) {
  if (levels) *levels = (h->launch && h->coop_prefix == 2) ? h->plan.levels : 0;
  for (int l = 0; l < 3; ++l) if (groups) groups[l] = h->plan.g[l];
  for (int l = 0; l < 4; ++l) if (counts) counts[l] = h->plan.n[l];
  return CLR_OK;
}

int clr_batch_debug_get_starts(clr_batch* h, double* starts) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->launch || !starts) return fail(CLR_INVALID_ARGUMENT, "start states are kept for widths 1..8");
  const size_t n = (size_t)h->B * h->nchunk * h->launch->start_doubles;
  HIP_TRY(hipMemcpyAsync(starts, h->starts.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CLR_OK;
}

int clr_batch_debug_compose_check(clr_batch* h, int group, double* max_abs_diff, double* max_abs_value) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->launch || group < 2 || !h->evaluated)
    return fail(CLR_INVALID_ARGUMENT, "compose check: widths 1..8, group >= 2, after an evaluation");
  clr::BatchParams P;
  if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
  const size_t np = (size_t)(h->nchunk + group - 1) / group, E = (size_t)h->launch->elem_doubles;
  const size_t n = (size_t)h->B * np * E;
  DevBuf a, b;
  if ((st = a.reserve(n)) != CLR_OK || (st = b.reserve(n)) != CLR_OK) return st;
  h->launch->compose_check(P, group, a.p, b.p, h->stream);
  HIP_TRY(hipGetLastError());
  std::vector<double> ha(n), hb(n);
  HIP_TRY(hipMemcpyAsync(ha.data(), a.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(hb.data(), b.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  a.release();
  b.release();
  // per element (one composed group) and per block of it (A | b | C | eta | Jm): the largest difference against the
  // block's largest magnitude; the last group of a problem may hold the padded last chunk (never applied): skipped
  const int J = h->J, SZ = J * (J + 1) / 2;
  const size_t off[6] = {0, (size_t)J * J, (size_t)J * J + J, (size_t)J * J + J + SZ, (size_t)J * J + 2 * J + SZ, E};
  double worst = 0.0, big = 0.0;
  for (size_t e = 0; e < (size_t)h->B * np; ++e) {
    if (e % np == np - 1) continue;
    for (int blk = 0; blk < 5; ++blk) {
      double d = 0.0, m = 0.0;
      for (size_t i = off[blk]; i < off[blk + 1]; ++i) {
        const double x = ha[e * E + i], y = hb[e * E + i];
        if (!(fabs(x - y) <= d)) d = fabs(x - y);
        if (!(fabs(y) <= m)) m = fabs(y);
      }
      const double r = m > 0.0 ? d / m : d;
      if (!(r <= worst)) worst = r;
      if (!(m <= big)) big = m;
    }
  }
  if (max_abs_diff) *max_abs_diff = worst;
  if (max_abs_value) *max_abs_value = big;
  return CLR_OK;
}

int clr_batch_set_summarize_mode(clr_batch* h, int mode) {
  if (mode < -1 || mode > 2) return fail(CLR_INVALID_ARGUMENT, "summarize mode must be -1, 0, 1 or 2");
  if (mode != h->summarize_mode) h->relayout_pending = true;
  h->summarize_mode = mode;
  if (mode >= 0) h->pipeline_pinned = true;
  return CLR_OK;
}

int clr_batch_get_summarize_kernel(const clr_batch* h, int* kind) {
  if (!kind) return fail(CLR_INVALID_ARGUMENT, "kind is null");
  *kind = split_active(h) ? ((h->summarize_mode != 1 && lazy_eligible(h)) ? 2 : 1) : 0;
  if (!h->launch)  // wide plans: plain or lazy flavour of the one-wave-per-chunk summarize
    *kind = (h->nchunk > 1 && (h->summarize_mode < 0 || h->summarize_mode == 2) && lazy_eligible(h)) ? 2 : 0;
  return CLR_OK;
}

int clr_batch_set_general(clr_batch* h, int J_general, const double* A, long A_stride, const double* U, long U_stride,
                          const double* V, long V_stride) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (J_general < 0) return fail(CLR_INVALID_ARGUMENT, "J_general must be >= 0");
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  if (J_general == 0) {  // back to the celerite-terms-only plan
    h->J_general = 0;
    return CLR_OK;
  }
  if (!A || !U || !V) return fail(CLR_INVALID_ARGUMENT, "general terms need A, U and V");
  if (h->J + J_general > CLR_MAX_WIDTH) return fail(CLR_UNSUPPORTED, "width above CLR_MAX_WIDTH");
  const long N = h->N, UV = (long)J_general * N;
  if ((A_stride != 0 && A_stride != N) || (U_stride != 0 && U_stride != UV) || (V_stride != 0 && V_stride != UV))
    return fail(CLR_INVALID_ARGUMENT, "general-term strides must be 0 (shared) or the size of one problem's block");
  auto count = [&](long sd, long one) { return (size_t)(sd == 0 ? one : one * (long)h->B); };
  if ((st = upload(h->gA, A, count(A_stride, N), h->stream)) != CLR_OK) return st;
  if ((st = upload(h->gU, U, count(U_stride, UV), h->stream)) != CLR_OK) return st;
  if ((st = upload(h->gV, V, count(V_stride, UV), h->stream)) != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->J_general = J_general;
  h->gA_stride = A_stride; h->gU_stride = U_stride; h->gV_stride = V_stride;
  // Total widths up to 64 run on the wave-per-(problem, chunk) kernels (wide_kernels.hip, GEN flavour): the general rows
  // are one more row class there.  Chunks as for any wide plan: one round of two waves per SIMD, the scan up to width 32.
  h->gen_nchunk = 0;
  const int Wt = h->J + J_general;
  if (Wt <= clr::wide_max_width()) {
    int nchunk = (Wt <= clr::wide_scan_max_width() && h->B <= 1024) ? 2048 / h->B : 1;
    if (nchunk > 16) nchunk = 16;
    while (nchunk > 1 && h->N / nchunk < 512) --nchunk;
    if (h->warm_explicit_chunks > 0 && Wt <= clr::wide_scan_max_width())  // (an explicit clr_batch_set_chunks is honoured here too)
      nchunk = std::min(h->warm_explicit_chunks, std::max(1, h->N / 64));
    if (nchunk < 1) nchunk = 1;
    int L = (h->N + nchunk - 1) / nchunk;
    if (nchunk > 1) L = (L + 7) & ~7;
    nchunk = (h->N + L - 1) / L;
    int L0 = 0;
    if (nchunk > 1 && h->wide_first_ratio > 1.0) {  // (the riderless, longer first chunk: clr_batch_set_chunks)
      int L2 = (int)ceil(h->N / (nchunk - 1 + h->wide_first_ratio));
      L2 = (L2 + 7) & ~7;
      const long first = (long)h->N - (long)(nchunk - 1) * L2;
      if (L2 >= 64 && first >= L2) { L = L2; L0 = (int)first; }
    }
    const size_t pc = (size_t)h->B * nchunk, JP = Wt <= 16 ? 16 : 32, SZ = JP * (JP + 1) / 2;
    if (nchunk > 1) {
      if ((st = h->gen_elems.reserve(pc * (JP * JP + JP + SZ + JP + SZ))) != CLR_OK) return st;
      if ((st = h->gen_starts.reserve(pc * (SZ + JP))) != CLR_OK) return st;
    }
    if ((st = h->gen_part.reserve(pc * 4)) != CLR_OK) return st;
    if ((st = h->gen_cond.reserve(pc * 4)) != CLR_OK) return st;
    if (h->gen_flags) (void)hipFree(h->gen_flags);
    h->gen_flags = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->gen_flags), (2 * pc + (size_t)h->B) * sizeof(int)));
    HIP_TRY(hipMemsetAsync(h->gen_flags, 0, (2 * pc + (size_t)h->B) * sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(h->gen_cond.p, 0, pc * 4 * sizeof(double), h->stream));
    h->gen_nchunk = nchunk; h->gen_L = L; h->gen_L0 = L0;
  }
  return CLR_OK;
}

static bool warm_runs(const clr_batch* h, int materialize);
static bool small_runs(const clr_batch* h, int materialize);

int clr_batch_set_general_route(clr_batch* h, int route) {
  if (route != -1 && route != 1) return fail(CLR_INVALID_ARGUMENT, "general route: -1 (automatic) or 1 (sequential kernel)");
  h->general_route = route;
  return CLR_OK;
}

int clr_batch_set_small_mode(clr_batch* h, int mode) {
  if (mode < -1 || mode > 1) return fail(CLR_INVALID_ARGUMENT, "small mode: -1 (auto), 0 (off) or 1 (whenever supported)");
  int st = warm_resolve(h, nullptr);
  if (st != CLR_OK) return st;
  h->small_mode = mode;
  return CLR_OK;
}

int clr_batch_get_small_mode(const clr_batch* h, int* active) {
  if (active) *active = (!warm_runs(h, 0) && small_runs(h, 0)) ? 1 : 0;
  return CLR_OK;
}

int clr_batch_set_warm_start(clr_batch* h, int mode, int forced_warmup) {
  if (mode < -1 || mode > 1 || (mode == 1 && forced_warmup < 1))
    return fail(CLR_INVALID_ARGUMENT, "warm start: mode -1 (auto), 0 (off) or 1 (forced, with a warm-up length >= 1)");
  int st = warm_resolve(h, nullptr);
  if (st != CLR_OK) return st;
  h->warm_mode = mode;
  h->warm_forced_K = forced_warmup;
  h->warm_boost = 0;
  h->warm_active = false;  // (decided by the next clr_batch_set_coefficients)
  h->have_coeffs = false;
  return CLR_OK;
}

int clr_batch_get_warm_start(const clr_batch* h, int* active, int* nchunk, int* chunk_len, int* warmup_min,
                             int* warmup_max, int* settled, int* fallbacks) {
  if (active) *active = h->warm_active ? 1 : 0;
  if (nchunk) *nchunk = h->wnchunk;
  if (chunk_len) *chunk_len = h->wL;
  int lo = 0, hi = 0;
  if (h->warm_active)
    for (int k : h->warm_K) {
      if (k <= 0) continue;
      lo = (lo == 0 || k < lo) ? k : lo;
      hi = k > hi ? k : hi;
    }
  if (warmup_min) *warmup_min = lo;
  if (warmup_max) *warmup_max = hi;
  if (settled) *settled = h->warm_settled;
  if (fallbacks) *fallbacks = h->warm_fallbacks;
  return CLR_OK;
}

int clr_batch_set_replay_source(clr_batch* h, int source) {
  if (source < -1 || source > 1) return fail(CLR_INVALID_ARGUMENT, "replay source must be -1, 0 or 1");
  h->replay_source = source;
  h->pipeline_pinned = true;
  return CLR_OK;
}

int clr_batch_set_library_trig(clr_batch* h, int force) {
  h->force_library_trig = force ? 1 : 0;
  return CLR_OK;
}

int clr_batch_set_layout(clr_batch* h, int layout) {
  if (layout < 0 || layout > 2) return fail(CLR_INVALID_ARGUMENT, "layout must be 0, 1 or 2");
  h->layout = layout;
  h->relayout_pending = true;
  h->pipeline_pinned = true;
  return CLR_OK;
}

// wide path.  One chunk: the sequential sweep (one wave per problem).  Several chunks: summarize ->
// prefix -> correct (+ conditioning decision) -> finalize from the chunk summaries.  Forced-exact runs
// and the problems whose conditioning record is above the bound replay every chunk from its scanned
// start state and check the end states against the scan.  Problems the certificate flagged, or whose
// replay did not meet the scan, are then walked by the sequential sweep itself (one wave per such
// problem over all N samples), which overwrites their results: nothing of theirs depends on the scan.
static void wide_flow(clr::BatchParams& P, int J_real, int J_comp, hipStream_t stream, hipEvent_t* ev) {
  auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], stream); };
  const int JP = J_real + 2 * J_comp + P.J_general <= 16 ? 16 : 32;
  mark(1);
  if (P.nchunk > 1) clr::launch_wide_summarize(P, J_real, J_comp, stream);
  mark(2);
  clr::launch_wide_prefix(P, JP, stream);
  mark(3);
  clr::launch_wide_correct(P, JP, stream);
  mark(4);
  // one chunk: the sweep itself; several: the chunked replay of forced runs and of the problems the
  // conditioning record marked (level 1), with its end states checked against the scan
  clr::launch_wide_loglike(P, J_real, J_comp, stream);
  if (P.nchunk > 1) {
    clr::launch_wide_check_replay(P, stream);
    clr::launch_finalize(P, stream);
    clr::BatchParams S = P;  // the flagged problems, sequentially
    S.nchunk = 1; S.L = P.N; S.L0 = 0; S.seq_only = 1; S.force_exact = 1;
    clr::launch_wide_loglike(S, J_real, J_comp, stream);
  }
  mark(5);
  mark(6);
}
static void wide_launch(clr_batch* h, clr::BatchParams& P, hipEvent_t* ev) {
  wide_flow(P, h->J_real, h->J_comp, h->stream, ev);
}

static const int PROF_NK = 6, PROF_MAX_STEPS = 4096;

int clr_batch_set_profiling(clr_batch* h, int on) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  h->prof_on = (on == 2 && h->launch) ? 2 : (on ? 1 : 0);  // 2: only the summarize (dominant) kernel is bracketed (widths 1..8)
  h->prof_steps = 0;
  return CLR_OK;
}

int clr_batch_get_profile(clr_batch* h, double* kernel_ms /* [6] */, int* steps) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(h->stream));
  double k[PROF_NK] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < h->prof_steps; ++i)
    for (int j = 0; j < PROF_NK; ++j) {
      if (h->prof_on == 2 && j != 1) continue;  // (only events 1 and 2 were recorded)
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, h->prof_events[(size_t)i * (PROF_NK + 1) + j],
                                  h->prof_events[(size_t)i * (PROF_NK + 1) + j + 1]));
      k[j] += ms;
    }
  if (kernel_ms)
    for (int j = 0; j < PROF_NK; ++j) kernel_ms[j] = k[j];
  if (steps) *steps = h->prof_steps;
  return CLR_OK;
}

// a batch of short, narrow problems: the whole fused evaluation in ONE launch, one workgroup per problem
// (small_batch_kernel, small_kernels.hip); problems it cannot certify stay pending for the scan pipeline
static bool small_runs(const clr_batch* h, int materialize) {
  if (h->grad_scan_only || h->small_mode == 0) return false;
  if (!h->launch || materialize || h->force_exact || !h->wints || h->J_general > 0) return false;
  if (!clr::small_batch_supported(h->J_real, h->J_comp, h->N)) return false;
  // (automatic: while a workgroup per problem still fits one round of the chip -- above that the scan pipeline's
  //  throughput wins, profiles/r04i_small_batch.txt)
  if (h->small_mode == 1) return true;
  // automatic: not when the caller tuned the scan pipeline explicitly, and only while a workgroup per problem fits
  // one round of the chip (widths 3, 4: one workgroup per CU by registers and LDS; narrower: four) -- above that the
  // pipeline's throughput wins (profiles/r04i_small_batch.txt: 1024 x 1e4 x width 4 0.29 ms against 0.34)
  return !h->pipeline_pinned && h->B <= (h->J >= 3 ? 256 : 1024);
}

// a plan with general terms on the wide kernels: its own chunking and workspace (clr_batch_set_general)
static void general_wide_params(const clr_batch* h, const clr::BatchParams& P, clr::BatchParams& W) {
  W = P;
  const size_t pc = (size_t)h->B * h->gen_nchunk;
  W.nchunk = h->gen_nchunk; W.L = h->gen_L; W.L0 = h->gen_L0;
  W.t = h->t.p; W.diag = h->diag.p; W.y = h->y.p;
  W.t_stride = h->t_stride; W.diag_stride = h->diag_stride; W.y_stride = h->y_stride;
  W.lane_is = 1; W.lane_cs = W.L; W.staged = 0; W.split = 0; W.only_pending = 0;
  W.split_lazy = ((h->summarize_mode < 0 || h->summarize_mode == 2) && W.nchunk > 1 && lazy_eligible(h)) ? 1 : 0;
  W.coop_prefix = 1;
  W.J_general = h->J_general;
  W.gen_A = h->gA.p; W.gen_U = h->gU.p; W.gen_V = h->gV.p;
  W.gen_A_stride = h->gA_stride; W.gen_U_stride = h->gU_stride; W.gen_V_stride = h->gV_stride;
  W.elems = h->gen_elems.p; W.starts = h->gen_starts.p;
  W.part = h->gen_part.p; W.partx = h->gen_part.p + pc * 2;
  W.cond = h->gen_cond.p; W.egerr = h->gen_cond.p + pc * 3;
  W.flags = h->gen_flags; W.flagsx = h->gen_flags + pc; W.need_exact = h->gen_flags + 2 * pc;
  W.force_exact = (h->force_exact || W.nchunk < 2) ? 1 : 0;
  W.wide_materialize = 0;
}

static bool warm_runs(const clr_batch* h, int materialize) {
  if (h->grad_scan_only) return false;
  return h->launch && h->warm_active && !materialize && !h->force_exact && h->nchunk > 1 && h->wnchunk > 1;
}

// the warm kernel's copy of the series, (re)built when the series or the warm chunking changed
static int warm_copy(clr_batch* h) {
  if (!h->warm_copy_pending) return CLR_OK;
  const size_t cells = (size_t)h->wrows * h->wnchunk;
  auto nsrc = [&](long sd) { return (size_t)(sd == 0 ? 1 : h->B); };
  int st;
  if ((st = h->wT.reserve(nsrc(h->t_stride) * cells)) != CLR_OK) return st;
  if ((st = h->wD.reserve(nsrc(h->diag_stride) * cells)) != CLR_OK) return st;
  if ((st = h->wY.reserve(nsrc(h->y_stride) * cells)) != CLR_OK) return st;
  struct { DevBuf* src; DevBuf* dst; long stride; int pad; } jobs[3] = {
      {&h->t, &h->wT, h->t_stride, 1}, {&h->diag, &h->wD, h->diag_stride, 2}, {&h->y, &h->wY, h->y_stride, 0}};
  for (auto& j : jobs)
    clr::launch_relayout_warm(j.src->p, j.stride, j.dst->p, j.stride ? (long)cells : 0, j.stride ? h->B : 1, h->N,
                              h->wL, h->wnchunk, h->wKpad, h->wrows, j.pad, h->stream);
  h->warm_copy_pending = false;
  return CLR_OK;
}

// K per problem on the device, when set_series re-selected it after the coefficients were uploaded
static int warm_upload_K(clr_batch* h) {
  if (!h->warm_K_dirty || !h->wints) return CLR_OK;
  HIP_TRY(hipMemcpyAsync(h->wints + (size_t)h->B * h->wnchunk + h->B, h->warm_K.data(), (size_t)h->B * sizeof(int),
                         hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));  // (pageable source)
  h->warm_K_dirty = false;
  return CLR_OK;
}

// the scan pipeline for the problems the warm path left pending (single-wave summarize on the row-major arrays)
static int warm_fallback(clr_batch* h) {
  clr::BatchParams P;
  h->in_fallback = true;
  int st = batch_params(h, 0, P);
  h->in_fallback = false;
  if (st != CLR_OK) return st;
  h->launch->summarize(P, h->stream);
  h->launch->prefix(P, h->stream);
  h->launch->correct(P, h->stream);
  h->launch->replay(P, 0, h->stream);
  h->launch->sequential(P, 0, h->stream);
  clr::launch_finalize(P, h->stream);
  HIP_TRY(hipGetLastError());
  return CLR_OK;
}

int clr_batch_enqueue(clr_batch* h, int materialize) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  clr::BatchParams P;
  if ((st = batch_params(h, materialize, P)) != CLR_OK) return st;
  // profiling: one event per kernel boundary of this evaluation, on the plan's stream
  hipEvent_t* ev = nullptr;
  if (h->prof_on && h->prof_steps < PROF_MAX_STEPS) {
    const size_t need = (size_t)(h->prof_steps + 1) * (PROF_NK + 1);
    while (h->prof_events.size() < need) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      h->prof_events.push_back(e);
    }
    ev = &h->prof_events[(size_t)h->prof_steps * (PROF_NK + 1)];
    ++h->prof_steps;
  }
  const bool all_marks = h->prof_on != 2;
  auto mark = [&](int i) { if (ev && (all_marks || i == 1 || i == 2)) (void)hipEventRecord(ev[i], h->stream); };
  h->evaluated = true;
  if (h->J_general > 0) {  // general terms: the any-width sequential recurrence, one workgroup per problem
    if (materialize) return fail(CLR_UNSUPPORTED, "materialising runs with general terms: use CholeskySolver");
    h->warm_inflight = false;
    if (h->gen_nchunk > 0 && h->general_route != 1) {
      // the wide kernels with the general rows as a third row class (chunked scan up to total width 32)
      clr::BatchParams W;
      general_wide_params(h, P, W);
      mark(0);
      wide_flow(W, h->J_real, h->J_comp, h->stream, ev);
      HIP_TRY(hipGetLastError());
      return CLR_OK;
    }
    clr::GenericBatch G;
    memset(&G, 0, sizeof(G));
    G.B = h->B; G.N = h->N; G.J_real = h->J_real; G.J_comp = h->J_comp; G.J_general = h->J_general;
    G.a_real = P.a_real; G.c_real = P.c_real; G.a_comp = P.a_comp; G.b_comp = P.b_comp; G.c_comp = P.c_comp;
    G.d_comp = P.d_comp; G.jitter = P.jitter;
    G.t = h->t.p; G.diag = h->diag.p; G.y = h->y.p;
    G.t_stride = h->t_stride; G.diag_stride = h->diag_stride; G.y_stride = h->y_stride;
    G.A = h->gA.p; G.U = h->gU.p; G.V = h->gV.p;
    G.A_stride = h->gA_stride; G.U_stride = h->gU_stride; G.V_stride = h->gV_stride;
    G.out_ll = P.out_ll; G.out_logdet = P.out_logdet; G.out_quad = P.out_quad; G.out_status = P.out_status;
    mark(0); mark(1);
    clr::launch_generic_loglike_batch(G, h->stream);
    mark(2); mark(3); mark(4); mark(5); mark(6);
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  if (!h->launch) {
    mark(0);
    wide_launch(h, P, ev);
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  mark(0);
  h->warm_inflight = false;
  if (!warm_runs(h, materialize) && small_runs(h, materialize)) {
    clr::BatchParams Sp;
    h->in_fallback = true;  // (the row-major arrays)
    st = batch_params(h, 0, Sp);
    h->in_fallback = false;
    if (st != CLR_OK) return st;
    mark(1);
    clr::launch_small_batch(h->J_real, h->J_comp, Sp, 256, h->stream);
    mark(2); mark(3); mark(4); mark(5); mark(6);
    h->warm_inflight = true;  // (pending problems are settled like the warm path's: warm_resolve)
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  if (warm_runs(h, materialize)) {
    // series that forget: the plain recurrence per chunk with a warm-up + the boundary check; problems it cannot
    // settle are marked pending and go through the scan pipeline when the results are asked for
    clr::BatchParams Wp;
    if ((st = warm_copy(h)) != CLR_OK) return st;
    if ((st = warm_upload_K(h)) != CLR_OK) return st;
    h->in_fallback = true;  // (the row-major arrays, no role split)
    st = batch_params(h, 0, Wp);
    h->in_fallback = false;
    if (st != CLR_OK) return st;
    mark(1);
    h->launch->warm(Wp, h->stream);
    mark(2); mark(3); mark(4); mark(5); mark(6);
    h->warm_inflight = true;
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  if (h->relayout_pending && batch_relayout(h)) h->relayout_pending = false;
  mark(1);
  h->launch->summarize(P, h->stream);
  mark(2);
  h->launch->prefix(P, h->stream);
  mark(3);
  h->launch->correct(P, h->stream);  // (also on forced-exact runs: flags + conditioning record)
  mark(4);
  h->launch->replay(replay_view(h, P, materialize), materialize ? 2 : 0, h->stream);  // forced-exact / materialising runs only
  h->launch->sequential(P, materialize ? 2 : 0, h->stream);  // flagged / ill-conditioned problems only
  mark(5);
  clr::launch_finalize(P, h->stream);
  mark(6);
  // (capturing these five launches in a hipGraph was measured: no gain -- the gaps between
  //  dependent kernels are on the device side; profiles/r01r_small_batches.log)
  HIP_TRY(hipGetLastError());
  return CLR_OK;
}

int clr_batch_fp32_probe(clr_batch* h, double* logdet, double* quad, double* ms) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (h->launch || h->J > clr::wide_f32_probe_max_width())
    return fail(CLR_UNSUPPORTED, "the fp32 probe covers widths 9..32");
  clr::BatchParams P;
  if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
  const size_t B = (size_t)h->B;
  DevBuf tmp;
  if ((st = tmp.reserve(2 * B)) != CLR_OK) return st;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  clr::launch_wide_f32_probe(P, h->J_real, h->J_comp, tmp.p, tmp.p + B, h->stream);  // warm-up
  HIP_TRY(hipEventRecord(e0, h->stream));
  clr::launch_wide_f32_probe(P, h->J_real, h->J_comp, tmp.p, tmp.p + B, h->stream);
  HIP_TRY(hipEventRecord(e1, h->stream));
  HIP_TRY(hipGetLastError());
  if (logdet) HIP_TRY(hipMemcpyAsync(logdet, tmp.p, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (quad) HIP_TRY(hipMemcpyAsync(quad, tmp.p + B, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, e0, e1));
  if (ms) *ms = t;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  tmp.release();
  return CLR_OK;
}

// Problems the warm path could not settle (boundary mismatch, flagged pivot, not eligible) carry a pending status
// until the scan pipeline has run for them.  That pipeline reads the plan's CURRENT coefficients, series and
// chunking, so it must run before any of them changes: every state-changing entry point, clr_batch_synchronize and
// clr_batch_get_results call this first.  *pin_current (may be null): the pinned staging buffer holds the final
// results (ll | logdet | quad | status) of this evaluation.
static int warm_resolve(clr_batch* h, bool* pin_current) {
  if (pin_current) *pin_current = false;
  if (!h->warm_inflight) return CLR_OK;
  h->warm_inflight = false;
  const size_t B = (size_t)h->B, words = 3 * B + (B + 1) / 2;
  int st;
  if ((st = reserve_pinned(h, words)) != CLR_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int* stw = reinterpret_cast<const int*>(h->pin + 3 * B);
  int pending = 0;
  for (size_t b = 0; b < B; ++b) pending += stw[b] == clr::CLR_PENDING_STATUS;
  h->warm_fallbacks = pending;
  h->warm_settled = (int)B - pending;
  if (pending) {
    if ((st = warm_fallback(h)) != CLR_OK) return st;
    HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // many mismatches among the problems that did warm up: longer warm-ups from the next coefficients on
    size_t eligible = 0;
    for (int k : h->warm_K) eligible += k > 0;
    const long failed = (long)pending - (long)(B - eligible);
    if (h->warm_mode < 0 && failed * 10 > (long)eligible && h->warm_boost < clr_batch::WARM_NK - 1) ++h->warm_boost;
  } else if (h->warm_mode < 0 && h->warm_boost > 0 && ++h->warm_clean >= 8) {
    --h->warm_boost;  // eight clean evaluations in a row: try the shorter warm-ups again
    h->warm_clean = 0;
  }
  if (pending) h->warm_clean = 0;
  if (pin_current) *pin_current = true;
  return CLR_OK;
}

int clr_batch_synchronize(clr_batch* h) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CLR_OK;
}

int clr_batch_get_results(clr_batch* h, double* loglike, double* logdet, double* quad,
                          int* status) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  const size_t B = (size_t)h->B, words = 3 * B + (B + 1) / 2;
  if ((st = reserve_pinned(h, words)) != CLR_OK) return st;
  // pending problems of a warm evaluation first (they leave the final results in the staging buffer); otherwise one
  // copy into the pinned staging buffer (ll | logdet | quad | status), then host memcpys
  bool pin_current = false;
  if ((st = warm_resolve(h, &pin_current)) != CLR_OK) return st;
  if (!pin_current) {
    HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (loglike) memcpy(loglike, h->pin, B * sizeof(double));
  if (logdet) memcpy(logdet, h->pin + B, B * sizeof(double));
  if (quad) memcpy(quad, h->pin + 2 * B, B * sizeof(double));
  if (status) memcpy(status, h->pin + 3 * B, B * sizeof(int));
  return CLR_OK;
}

// time spanned by every scan chunk (its samples and the move to the next chunk's first sample), one thread per chunk
__global__ void chunk_span_kernel(const double* t, long t_stride, int N, int L, int nchunk, int nsrc, double* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nsrc * nchunk) return;
  const long b = idx / nchunk, c = idx % nchunk;
  const long first = c * (long)L, last = std::min<long>((c + 1) * (long)L, N - 1);
  out[idx] = first < N ? t[b * t_stride + last] - t[b * t_stride + first] : 0.0;
}

// per problem (one entry for a shared series): the longest time any scan chunk spans; cached until the series or the
// chunking changes
static int grad_chunk_spans(clr_batch* h) {
  if (h->grad_span_valid) return CLR_OK;
  const int nsrc = h->t_stride == 0 ? 1 : h->B;
  const size_t n = (size_t)nsrc * h->nchunk;
  DevBuf tmp;
  int st = tmp.reserve(n);
  if (st != CLR_OK) return st;
  hipLaunchKernelGGL(chunk_span_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->t.p, h->t_stride,
                     h->N, h->L, h->nchunk, nsrc, tmp.p);
  std::vector<double> spans(n);
  const bool ok = hipGetLastError() == hipSuccess &&
                  hipMemcpyAsync(spans.data(), tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                  hipStreamSynchronize(h->stream) == hipSuccess;
  tmp.release();
  if (!ok) return fail(CLR_HIP_ERROR, "chunk span kernel failed");
  h->grad_span.assign(nsrc, 0.0);
  for (int b = 0; b < nsrc; ++b)
    for (int c = 0; c < h->nchunk; ++c) {
      const double v = spans[(size_t)b * h->nchunk + c];
      if (!(v <= h->grad_span[b])) h->grad_span[b] = v;  // (NaN sticks: as many slots as steps)
    }
  h->grad_span_valid = true;
  return CLR_OK;
}

// Value and gradient of every problem of the plan at the coefficients in force, parallel in n (clr_grad_core.h):
// the evaluation by the scan, then per chunk the riders and the tangents of every direction group from the scanned
// start states, then the walk over the chunks.  Problems the scan routed to the sequential recurrence take the
// sequential gradient kernel (grad_kernels.hip).
// The plan gradient at widths 9..32 and with general terms (total width <= 32): the wide scan's evaluation, then the
// riders of every chunk from its element, one tangent wave per (direction, chunk) from the scanned start states, and a
// walk over the chunks per direction (wide_grad_kernels.hip); problems the evaluation sent to the sequential recurrence
// take the sequential tangent kernel.
static int wide_batch_grad(clr_batch* h, double* value, double* grad, int* status) {
  const bool general = h->J_general > 0;
  const int Wt = h->J + h->J_general;
  if (general && (h->gen_nchunk < 2 || h->general_route == 1))
    return fail(CLR_UNSUPPORTED, "the plan gradient with general terms needs the chunked wide scan (total width <= 32, N >= 1024)");
  if (!general && (h->launch || h->nchunk < 2))
    return fail(CLR_UNSUPPORTED, "the plan gradient at widths 9..32 needs a chunked plan: use clr_batch_grad_log_likelihood");
  if (Wt > clr::wide_scan_max_width())
    return fail(CLR_UNSUPPORTED, "the plan gradient covers total widths up to 32: use clr_batch_grad_log_likelihood");
  int st = clr_batch_enqueue(h, 0);
  if (st != CLR_OK) return st;
  clr::BatchParams P0, P;
  if ((st = batch_params(h, 0, P0)) != CLR_OK) return st;
  if (general) general_wide_params(h, P0, P); else P = P0;
  const size_t B = (size_t)h->B, NG = 1 + 2 * (size_t)h->J_real + 4 * (size_t)h->J_comp;
  const int JP = Wt <= 16 ? 16 : 32;
  const size_t pc = B * (size_t)P.nchunk, RID = 2 * (size_t)JP * JP + JP, OUT = (size_t)JP * JP + JP + 2;
  if ((st = h->g_riders.reserve(pc * RID)) != CLR_OK) return st;
  if ((st = h->g_out.reserve(pc * NG * OUT)) != CLR_OK) return st;
  if ((st = h->g_res.reserve(B * (NG + 2))) != CLR_OK) return st;  // value | grad | status (ints in the last B doubles)
  double* d_value = h->g_res.p;
  double* d_grad = h->g_res.p + B;
  int* d_status = reinterpret_cast<int*>(h->g_res.p + B + B * NG);

  clr::WideGradWalk W;
  memset(&W, 0, sizeof(W));
  W.B = h->B; W.NG = (int)NG; W.nchunk = P.nchunk; W.JP = JP; W.N = h->N;
  W.elems = P.elems; W.starts = P.starts; W.riders = h->g_riders.p; W.rec = h->g_out.p;
  W.level = P.need_exact; W.ll = P.out_ll; W.ll_status = P.out_status; W.jitter = P.jitter;
  W.out_value = d_value; W.out_grad = d_grad; W.out_status = d_status;

  clr::GradParams G;
  memset(&G, 0, sizeof(G));
  G.N = h->N; G.J_real = h->J_real; G.J_comp = h->J_comp; G.J_general = h->J_general;
  G.a_real = P.a_real; G.c_real = P.c_real; G.a_comp = P.a_comp; G.b_comp = P.b_comp; G.c_comp = P.c_comp; G.d_comp = P.d_comp;
  G.jitter_b = P.jitter;
  if (general) {
    G.A = h->gA.p; G.U = h->gU.p; G.V = h->gV.p;
    G.A_stride = h->gA_stride; G.U_stride = h->gU_stride; G.V_stride = h->gV_stride;
  }
  G.t = h->t.p; G.diag = h->diag.p; G.y = h->y.p;
  G.t_stride = h->t_stride; G.diag_stride = h->diag_stride; G.y_stride = h->y_stride;
  G.fast_trig = P.fast_trig;
  G.B = h->B;
  G.only_level = P.need_exact;
  G.nchunk = P.nchunk; G.L = P.L; G.L0 = P.L0; G.JP = JP;
  G.starts = P.starts; G.rec = h->g_out.p;
  G.out_value = d_value; G.out_grad = d_grad; G.out_status = d_status;

  clr::launch_wide_grad_riders(W, h->stream);
  clr::launch_grad_chunked(G, h->stream);
  clr::launch_wide_grad_walk(W, h->stream);
  clr::launch_grad(G, h->stream);  // (sequential form: only the problems with level >= 2)
  HIP_TRY(hipGetLastError());
  std::vector<double> back(B * (NG + 2));
  HIP_TRY(hipMemcpyAsync(back.data(), h->g_res.p, back.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int* hst = reinterpret_cast<const int*>(back.data() + B + B * NG);
  int nfb = 0;
  std::vector<int> levels(B);
  HIP_TRY(hipMemcpy(levels.data(), P.need_exact, B * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; ++b) {
    nfb += levels[b] >= 2;
    const bool ok = hst[b] == CLR_OK;
    if (value) value[b] = ok ? back[b] : -INFINITY;
    if (status) status[b] = hst[b];
    if (grad)
      for (size_t g = 0; g < NG; ++g) grad[b * NG + g] = ok ? back[B + b * NG + g] : 0.0;
    if (ok && grad && !(h->host_jitter[b] > 2.220446049250313e-16)) grad[b * NG] = 0.0;  // solver.cpp:379-389 (sequential form too)
  }
  h->grad_fallbacks = nfb;
  h->grad_reverse_used = false;
  return CLR_OK;
}

int clr_batch_grad(clr_batch* h, double* value, double* grad, int* status) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->launch || h->J_general > 0) return wide_batch_grad(h, value, grad, status);
  const size_t B = (size_t)h->B, NG = 1 + 2 * (size_t)h->J_real + 4 * (size_t)h->J_comp, J = (size_t)h->J;
  const size_t SZ = J * (J + 1) / 2, OUT = SZ + J + 2, RID = J * J + J + SZ;
  h->grad_scan_only = true;
  st = clr_batch_enqueue(h, 0);
  h->grad_scan_only = false;
  if (st != CLR_OK) return st;
  clr::BatchParams P, Pi;
  h->in_fallback = true;  // (the row-major arrays)
  st = batch_params(h, 0, P);
  h->in_fallback = false;
  if (st != CLR_OK) return st;
  if ((st = batch_params(h, 0, Pi)) != CLR_OK) return st;  // (the evaluation's own view: interleaved copy if it has one)
  const bool scan_grad = P.fast_trig != 0;  // (only the fast-sincos flavour of the gradient kernels is built)
  // mode: reverse (one sweep for all partials, needs the per-sample record in HBM) unless asked otherwise or the
  // record does not fit; forward (one tangent per partial) as the fallback and the cross-check
  bool reverse = scan_grad && h->grad_mode != 1;
  const double w2 = (double)(J * J) / 64.0, groups = 1.0 + h->J_real + 2.0 * h->J_comp;
  auto choose_m = [&](bool rev) {
    // gradient chunks: m chunks of the scan each.  Modelled time: rounds of waves x steps per lane (forward: 2.7 us
    // per step of a tangent wave at width 8, ~ J^2, one wave per direction group; reverse: record + sweep, ~ 5 us)
    // + the walk over the gradient chunks
    int m = 1;
    double best = INFINITY;
    for (int k = 1; k <= h->nchunk; ++k) {
      const int ng = (h->nchunk + k - 1) / k;
      const double waves = (double)B * ((ng + 63) / 64) * (rev ? 1.0 : groups);
      const double step = rev ? 0.6 + 4.4 * w2 : 0.3 + 2.4 * w2;
      const double walk = rev ? 0.45 : 0.5 + 4.5 * w2 * J / 8.0;  // per chunk: a wave per problem / a thread per (problem, direction)
      const double tm = std::max(1.0, waves / 1024.0) * k * h->L * step + ng * walk;
      if (tm < best) { best = tm; m = k; }
    }
    return m;
  };
  auto set_chunks = [&](int m) {
    P.g_m = m;
    P.g_nchunk = (h->nchunk + m - 1) / m;
    if (m == 1 && Pi.lane_cs == 1 && !Pi.staged) {  // a gradient chunk is a scan chunk: read the interleaved copy
      P.t = Pi.t; P.diag = Pi.diag; P.y = Pi.y;
      P.t_stride = Pi.t_stride; P.diag_stride = Pi.diag_stride; P.y_stride = Pi.y_stride;
      P.lane_is = Pi.lane_is; P.lane_cs = Pi.lane_cs;
    } else {
      P.t = h->t.p; P.diag = h->diag.p; P.y = h->y.p;
      P.t_stride = h->t_stride; P.diag_stride = h->diag_stride; P.y_stride = h->y_stride;
      P.lane_is = 1; P.lane_cs = h->L;
    }
    return B * (size_t)P.g_nchunk;
  };
  if ((st = h->g_res.reserve(B * NG + B * (NG + 1) + B)) != CLR_OK) return st;  // result | fallback value, grad | fallback status
  P.g_res = h->g_res.p;
  h->grad_reverse_used = false;
  if (reverse) {
    const size_t pc = set_chunks(choose_m(true));
    const long Lg = (long)P.g_m * h->L;
    // stored states (GradStore, clr_grad_core.h): every grad_K steps when forced, else wherever the decay accumulated
    // since the last one reaches the growth budget -- sized from the problems' largest decay rates and the longest
    // time a chunk spans, with a factor 2 for the wave-wide trigger (a chunk that runs out of slots fails its
    // certificate and is redone in forward mode)
    long nalloc;
    if (h->grad_K > 0) {
      P.g_K = (int)std::min<long>(h->grad_K, Lg);
      nalloc = (Lg + P.g_K - 1) / P.g_K;
    } else {
      P.g_K = 0;
      if ((st = grad_chunk_spans(h)) != CLR_OK) return st;
      double need = 0.0;
      for (size_t b = 0; b < B; ++b) {
        const double v = h->host_cmax[b] * h->grad_span[h->t_stride == 0 ? 0 : b] * P.g_m / CLR_GRAD_GROWTH_BUDGET;
        if (!(v <= need)) need = v;
      }
      nalloc = (need == need && need < (double)Lg) ? (long)(2.0 * std::ceil(need)) + 8 : Lg;
      nalloc = std::min<long>(nalloc, Lg);
    }
    P.g_nalloc = (int)nalloc;
    P.g_rec_stride = Lg * (long)(J + 2) * P.g_nchunk;
    P.g_ck_stride = nalloc * (long)(SZ + J) * P.g_nchunk;
    const size_t nflag = B * (size_t)((P.g_nchunk + 63) / 64) * (size_t)Lg;
    bool flags_fit = true;  // (a flag buffer that does not fit degrades to forward mode like the record buffers)
    if (nflag > h->g_ckflag_cap) {
      if (h->g_ckflag) (void)hipFree(h->g_ckflag);
      h->g_ckflag = nullptr;
      h->g_ckflag_cap = 0;
      if (hipMalloc(reinterpret_cast<void**>(&h->g_ckflag), nflag) != hipSuccess) {
        h->g_ckflag = nullptr;
        flags_fit = false;
      } else {
        h->g_ckflag_cap = nflag;
      }
    }
    if (flags_fit) HIP_TRY(hipMemsetAsync(h->g_ckflag, 0, nflag, h->stream));
    P.g_ckflag = h->g_ckflag;
    const size_t small = pc * (RID + 3 * (SZ + J) + NG + 2) + B;
    if (!flags_fit || h->g_rec.reserve(B * (size_t)P.g_rec_stride) != CLR_OK ||
        h->g_ck.reserve(B * (size_t)P.g_ck_stride) != CLR_OK || h->g_riders.reserve(small) != CLR_OK) {
      h->g_rec.release(); h->g_ck.release();
      (void)hipGetLastError();
      reverse = false;  // (the record does not fit: one tangent per partial needs 50x less memory)
    } else {
      P.g_rec = h->g_rec.p; P.g_ck = h->g_ck.p;
      P.g_riders = h->g_riders.p;
      P.g_ends = P.g_riders + pc * RID;
      P.g_adj = P.g_ends + pc * (SZ + J);
      P.g_adj0 = P.g_adj + pc * (SZ + J);
      P.g_part = P.g_adj0 + pc * (SZ + J);
      P.g_drift = P.g_part + pc * NG;
      P.g_count = P.g_drift + pc;
      P.g_drift_max = P.g_count + pc;
      P.g_from_elems = (P.g_m == 1 && h->grad_riders_mode != 1) ? 1 : 0;
      h->launch->grad_reverse(P, h->stream);
      HIP_TRY(hipGetLastError());
      h->grad_reverse_used = true;
    }
  }
  auto forward_buffers = [&]() {
    const size_t pc = set_chunks(choose_m(false));
    int e;
    if ((e = h->g_riders.reserve(pc * RID)) != CLR_OK) return e;
    if ((e = h->g_out.reserve(pc * NG * OUT)) != CLR_OK) return e;
    P.g_riders = h->g_riders.p; P.g_out = h->g_out.p;
    P.g_rec = nullptr; P.g_ck = nullptr; P.g_ends = nullptr;
    return (int)CLR_OK;
  };
  if (scan_grad && !reverse) {
    if ((st = forward_buffers()) != CLR_OK) return st;
    h->launch->grad(P, h->stream);
    HIP_TRY(hipGetLastError());
  }
  std::vector<double> ll(B), ld(B), qd(B), res(B * NG);
  std::vector<int> stt(B), lvl(B);
  if ((st = clr_batch_get_results(h, ll.data(), ld.data(), qd.data(), stt.data())) != CLR_OK) return st;
  HIP_TRY(hipMemcpyAsync(res.data(), h->g_res.p, B * NG * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(lvl.data(), P.need_exact, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> drift;
  if (reverse) {
    drift.resize(B);
    HIP_TRY(hipMemcpyAsync(drift.data(), P.g_drift_max, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->grad_drift_max = 0.0;
  h->grad_forward_reruns = 0;
  if (reverse) {
    // the reverse sweep's certificates (one number per problem: the drift of its reconstructed states -- zero when
    // every state is stored -- and the mismatch between the adjoint a sweep arrives at and the one predicted from the
    // riders): problems beyond the tolerance are redone by the forward-mode kernels
    std::vector<int> mask(B, 0);
    int nre = 0;
    for (size_t b = 0; b < B; ++b) {
      if (stt[b] != CLR_OK || lvl[b] >= 2) continue;
      if (!(drift[b] <= h->grad_drift_max)) h->grad_drift_max = drift[b];
      if (!(drift[b] <= h->grad_drift_tol)) { mask[b] = 1; ++nre; }
    }
    h->grad_forward_reruns = nre;
    if (nre) {
      int* dmask = reinterpret_cast<int*>(h->g_res.p + B * NG + B * (NG + 1));
      HIP_TRY(hipMemcpyAsync(dmask, mask.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
      if ((st = forward_buffers()) != CLR_OK) return st;
      P.g_mask = dmask;
      // (the forward-mode result lands in the same g_res rows, only for the masked problems)
      h->launch->grad(P, h->stream);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(res.data(), h->g_res.p, B * NG * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      P.g_mask = nullptr;
    }
  }
  // the sequential gradient for the problems whose scanned start states are not certified
  int nfb = 0;
  for (size_t b = 0; b < B; ++b) nfb += (stt[b] == CLR_OK && (lvl[b] >= 2 || !scan_grad));
  h->grad_fallbacks = nfb;
  std::vector<double> fb;
  if (nfb) {
    clr::GradParams G;
    memset(&G, 0, sizeof(G));
    G.N = h->N; G.J_real = h->J_real; G.J_comp = h->J_comp; G.J_general = 0;
    G.a_real = P.a_real; G.c_real = P.c_real; G.a_comp = P.a_comp; G.b_comp = P.b_comp; G.c_comp = P.c_comp; G.d_comp = P.d_comp;
    G.jitter_b = P.jitter;
    G.t = h->t.p; G.diag = h->diag.p; G.y = h->y.p;
    G.t_stride = h->t_stride; G.diag_stride = h->diag_stride; G.y_stride = h->y_stride;
    G.B = h->B;
    G.fast_trig = P.fast_trig;
    G.only_level = scan_grad ? P.need_exact : nullptr;
    G.out_value = h->g_res.p + B * NG; G.out_grad = G.out_value + B;
    G.out_status = reinterpret_cast<int*>(G.out_grad + B * NG);
    clr::launch_grad(G, h->stream);
    HIP_TRY(hipGetLastError());
    fb.resize(B * NG);
    HIP_TRY(hipMemcpyAsync(fb.data(), G.out_grad, B * NG * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  const double cst = 3.14159265358979323846 * log((double)h->N);  // the reference's constant (solver.cpp:415)
  for (size_t b = 0; b < B; ++b) {
    const bool bad = stt[b] != CLR_OK;
    const bool from_fb = !bad && (lvl[b] >= 2 || !scan_grad);
    if (status) status[b] = stt[b];
    if (value) value[b] = bad ? -INFINITY : -0.5 * (qd[b] + ld[b] + cst);
    if (grad) {
      for (size_t g = 0; g < NG; ++g) grad[b * NG + g] = bad ? 0.0 : (from_fb ? fb[b * NG + g] : res[b * NG + g]);
      if (!(h->host_jitter[b] > 2.220446049250313e-16)) grad[b * NG] = 0.0;  // solver.cpp:379-389,419-426
    }
  }
  return CLR_OK;
}

int clr_batch_get_grad_fallbacks(const clr_batch* h, int* count) {
  if (!count) return fail(CLR_INVALID_ARGUMENT, "count is null");
  *count = h->grad_fallbacks;
  return CLR_OK;
}

int clr_batch_set_grad_mode(clr_batch* h, int mode, int stored_state_distance, double drift_tolerance) {
  if (mode < 0 || mode > 2 || stored_state_distance < 0) return fail(CLR_INVALID_ARGUMENT, "bad gradient mode");
  h->grad_riders_mode = mode == 2 ? 1 : 0;  // (2: reverse mode with the riders along the trajectory, for A/B runs)
  if (mode == 2) mode = 0;
  h->grad_mode = mode;
  h->grad_K = stored_state_distance;
  if (drift_tolerance > 0.0) h->grad_drift_tol = drift_tolerance;
  return CLR_OK;
}

int clr_batch_get_grad_info(const clr_batch* h, int* reverse_used, int* forward_reruns, double* drift_max) {
  if (reverse_used) *reverse_used = h->grad_reverse_used ? 1 : 0;
  if (forward_reruns) *forward_reruns = h->grad_forward_reruns;
  if (drift_max) *drift_max = h->grad_drift_max;
  return CLR_OK;
}

int clr_batch_get_factor(clr_batch* h, int p, double* phi, double* u, double* W, double* D) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->have_factor) return fail(CLR_NOT_COMPUTED, "no materialising run has been made");
  if (p < 0 || p >= h->B) return fail(CLR_INVALID_ARGUMENT, "problem index out of range");
  const size_t N = (size_t)h->N, J = (size_t)h->J, Nm1 = N - 1, cells = (size_t)h->L * h->nchunk;
  if (!h->launch) {  // widths 9..64: already in the reference's storage, problem after problem
    if (phi && J * Nm1) HIP_TRY(hipMemcpyAsync(phi, h->phi.p + p * J * Nm1, J * Nm1 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (u && J * Nm1) HIP_TRY(hipMemcpyAsync(u, h->u.p + p * J * Nm1, J * Nm1 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (W) HIP_TRY(hipMemcpyAsync(W, h->W.p + p * J * N, J * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (D) HIP_TRY(hipMemcpyAsync(D, h->D.p + p * N, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CLR_OK;
  }
  if ((st = h->fphi.reserve(J * Nm1)) != CLR_OK) return st;
  if ((st = h->fu.reserve(J * Nm1)) != CLR_OK) return st;
  if ((st = h->fW.reserve(J * N)) != CLR_OK) return st;
  if ((st = h->fD.reserve(N)) != CLR_OK) return st;
  clr::launch_deinterleave_factor(h->phi.p + p * J * cells, h->u.p + p * J * cells,
                                  h->W.p + p * J * cells, h->D.p + p * cells, h->fphi.p, h->fu.p,
                                  h->fW.p, h->fD.p, h->N, h->J, h->L, h->nchunk, h->stream);
  HIP_TRY(hipGetLastError());
  if (phi && J * Nm1) HIP_TRY(hipMemcpyAsync(phi, h->fphi.p, J * Nm1 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (u && J * Nm1) HIP_TRY(hipMemcpyAsync(u, h->fu.p, J * Nm1 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (W) HIP_TRY(hipMemcpyAsync(W, h->fW.p, J * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (D) HIP_TRY(hipMemcpyAsync(D, h->fD.p, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CLR_OK;
}

int clr_batch_run_timed(clr_batch* h, int materialize, int steps, int relayout_each_step,
                        double* total_ms, double* kernel_ms) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  clr::BatchParams P;
  if ((st = batch_params(h, materialize, P)) != CLR_OK) return st;
  if (steps < 1) steps = 1;
  h->evaluated = true;
  if (!warm_runs(h, materialize) && h->relayout_pending && !relayout_each_step && batch_relayout(h)) h->relayout_pending = false;
  // one event per kernel boundary per step, all recorded on the handle's stream
  const int NK = 6;
  std::vector<hipEvent_t> ev((size_t)steps * (NK + 1));
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < steps; ++i) {
    hipEvent_t* e = &ev[(size_t)i * (NK + 1)];
    HIP_TRY(hipEventRecord(e[0], h->stream));
    if (!h->launch) {  // wide path (one chunk: the whole sweep is reported in the "replay" slot)
      wide_launch(h, P, e);
      continue;
    }
    if (!warm_runs(h, materialize) && small_runs(h, materialize)) {  // (one launch, in the "summarize" slot)
      clr::BatchParams Sp;
      h->in_fallback = true;
      st = batch_params(h, 0, Sp);
      h->in_fallback = false;
      if (st != CLR_OK) return st;
      HIP_TRY(hipEventRecord(e[1], h->stream));
      clr::launch_small_batch(h->J_real, h->J_comp, Sp, 256, h->stream);
      for (int j = 2; j <= 6; ++j) HIP_TRY(hipEventRecord(e[j], h->stream));
      h->warm_inflight = true;
      continue;
    }
    if (warm_runs(h, materialize)) {  // (the warm path: recurrence + boundary check in the "summarize" slot)
      clr::BatchParams Wp;
      if (relayout_each_step) h->warm_copy_pending = true;  // (new series every step: the copy is rebuilt inside it)
      if ((st = warm_copy(h)) != CLR_OK) return st;
      if ((st = warm_upload_K(h)) != CLR_OK) return st;
      h->in_fallback = true;
      st = batch_params(h, 0, Wp);
      h->in_fallback = false;
      if (st != CLR_OK) return st;
      HIP_TRY(hipEventRecord(e[1], h->stream));
      h->launch->warm(Wp, h->stream);
      for (int j = 2; j <= 6; ++j) HIP_TRY(hipEventRecord(e[j], h->stream));
      h->warm_inflight = true;
      continue;
    }
    if (relayout_each_step) batch_relayout(h);
    HIP_TRY(hipEventRecord(e[1], h->stream));
    h->launch->summarize(P, h->stream);
    HIP_TRY(hipEventRecord(e[2], h->stream));
    h->launch->prefix(P, h->stream);
    HIP_TRY(hipEventRecord(e[3], h->stream));
    h->launch->correct(P, h->stream);
    HIP_TRY(hipEventRecord(e[4], h->stream));
    h->launch->replay(replay_view(h, P, materialize), materialize ? 2 : 0, h->stream);
    h->launch->sequential(P, materialize ? 2 : 0, h->stream);
    HIP_TRY(hipEventRecord(e[5], h->stream));
    clr::launch_finalize(P, h->stream);
    HIP_TRY(hipEventRecord(e[6], h->stream));
  }
  if (relayout_each_step && !warm_runs(h, materialize) && (h->layout == 1 || split_active(h)) && h->nchunk > 1)
    h->relayout_pending = false;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  double k[NK] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < steps; ++i) {
    hipEvent_t* e = &ev[(size_t)i * (NK + 1)];
    for (int j = 0; j < NK; ++j) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e[j], e[j + 1]));
      k[j] += ms;
    }
  }
  float tot = 0.f;
  HIP_TRY(hipEventElapsedTime(&tot, ev.front(), ev.back()));
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (total_ms) *total_ms = tot;
  if (kernel_ms)
    for (int j = 0; j < NK; ++j) kernel_ms[j] = k[j];
  return CLR_OK;
}

int clr_batch_grad_log_likelihood(int B, int N, int J_real, int J_comp, const double* jitter,
                                  const double* a_real, const double* c_real, const double* a_comp,
                                  const double* b_comp, const double* c_comp, const double* d_comp,
                                  const double* t, long t_stride, const double* diag, long diag_stride,
                                  const double* y, long y_stride, double* value, double* grad, int* status,
                                  int device) {
  if (B < 1 || N < 1 || J_real < 0 || J_comp < 0) return fail(CLR_INVALID_ARGUMENT, "bad sizes");
  if (J_real + 2 * J_comp < 1 || J_real + 2 * J_comp > 64) return fail(CLR_UNSUPPORTED, "widths 1..64");
  for (long sd : {t_stride, diag_stride, y_stride})
    if (sd != 0 && sd != N) return fail(CLR_INVALID_ARGUMENT, "series stride must be 0 (shared) or N");
  int st = require_device(device);
  if (st != CLR_OK) return st;
  if (J_real + 2 * J_comp <= 8 && N >= 512 && !getenv("CLR_GRAD_SEQUENTIAL")) {
    // widths 1..8: parallel in n through a plan (clr_batch_grad); short series and the other widths below
    clr_batch* h = clr_batch_create(B, N, J_real, J_comp, device);
    if (h) {
      st = clr_batch_set_series(h, t, t_stride, diag, diag_stride, y, y_stride);
      if (st == CLR_OK) st = clr_batch_set_coefficients(h, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp);
      if (st == CLR_OK) st = clr_batch_grad(h, value, grad, status);
      clr_batch_destroy(h);
      return st;
    }
  }
  const size_t Bn = (size_t)B, nr = Bn * J_real, nc = Bn * J_comp, NG = 1 + 2 * (size_t)J_real + 4 * (size_t)J_comp;
  auto count = [&](long sd) { return (size_t)(sd == 0 ? N : (long)N * B); };
  hipStream_t stream;
  HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  DevBuf buf, out;
  int* dstatus = nullptr;
  auto done = [&](int code) {
    buf.release(); out.release();
    if (dstatus) (void)hipFree(dstatus);
    (void)hipStreamDestroy(stream);
    return code;
  };
  // one staging vector: coefficients | jitter | t | diag | y
  std::vector<double> host;
  auto put = [&](const double* p, size_t n) { const size_t at = host.size(); if (n) host.insert(host.end(), p, p + n); return at; };
  const size_t o_ar = put(a_real, nr), o_cr = put(c_real, nr), o_ac = put(a_comp, nc), o_bc = put(b_comp, nc),
               o_cc = put(c_comp, nc), o_dc = put(d_comp, nc), o_j = put(jitter, Bn);
  const size_t o_t = put(t, count(t_stride)), o_d = put(diag, count(diag_stride)), o_y = put(y, count(y_stride));
  if ((st = upload(buf, host.data(), host.size(), stream)) != CLR_OK) return done(st);
  if ((st = out.reserve(Bn * (NG + 1))) != CLR_OK) return done(st);
  if (hipMalloc(reinterpret_cast<void**>(&dstatus), Bn * sizeof(int)) != hipSuccess) return done(fail(CLR_HIP_ERROR, "hipMalloc failed"));
  clr::GradParams P;
  memset(&P, 0, sizeof(P));
  const double* base = buf.p;
  P.N = N; P.J_real = J_real; P.J_comp = J_comp; P.J_general = 0;
  P.a_real = base + o_ar; P.c_real = base + o_cr; P.a_comp = base + o_ac; P.b_comp = base + o_bc;
  P.c_comp = base + o_cc; P.d_comp = base + o_dc;
  P.jitter_b = base + o_j;
  P.t = base + o_t; P.diag = base + o_d; P.y = base + o_y;
  P.t_stride = t_stride; P.diag_stride = diag_stride; P.y_stride = y_stride;
  P.B = B;
  {
    double dmax = 0.0;
    for (size_t i = 0; i < nc; ++i) { const double m = fabs(d_comp[i]); if (!(m <= dmax)) dmax = m; }
    P.fast_trig = (dmax * max_abs(t, (long)count(t_stride)) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
  }
  P.out_value = out.p; P.out_grad = out.p + Bn; P.out_status = dstatus;
  clr::launch_grad(P, stream);
  if (hipGetLastError() != hipSuccess) return done(fail(CLR_HIP_ERROR, "grad kernel launch failed"));
  std::vector<double> back(Bn * (NG + 1));
  std::vector<int> hst(Bn);
  if (hipMemcpyAsync(back.data(), out.p, back.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(hst.data(), dstatus, Bn * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return done(fail(CLR_HIP_ERROR, "copy back failed"));
  for (size_t b = 0; b < Bn; ++b) {
    const bool bad = hst[b] != CLR_OK;
    if (status) status[b] = hst[b];
    if (value) value[b] = bad ? -INFINITY : back[b];
    if (grad)
      for (size_t g = 0; g < NG; ++g) grad[b * NG + g] = bad ? 0.0 : back[Bn + b * NG + g];
    if (grad && !(jitter[b] > 2.220446049250313e-16)) grad[b * NG] = 0.0;  // solver.cpp:379-389,419-426
  }
  return done(CLR_OK);
}

int clr_batch_log_likelihood(int B, int N, int J_real, int J_comp, const double* jitter,
                             const double* a_real, const double* c_real, const double* a_comp,
                             const double* b_comp, const double* c_comp, const double* d_comp,
                             const double* t, long t_stride, const double* diag,
                             long diag_stride, const double* y, long y_stride, double* loglike,
                             double* logdet, double* quad, int* status, int device) {
  clr_batch* h = clr_batch_create(B, N, J_real, J_comp, device);
  if (!h) {
    // clr_batch_create recorded why (message in clr_last_error)
    if (B < 1 || N < 1 || J_real < 0 || J_comp < 0) return CLR_INVALID_ARGUMENT;
    const int width = J_real + 2 * J_comp;
    if (width < 1 || width > clr::wide_max_width()) return CLR_UNSUPPORTED;
    return visible_gfx950() > 0 ? CLR_HIP_ERROR : CLR_NO_DEVICE;
  }
  int st = clr_batch_set_series(h, t, t_stride, diag, diag_stride, y, y_stride);
  if (st == CLR_OK)
    st = clr_batch_set_coefficients(h, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp);
  if (st == CLR_OK) st = clr_batch_enqueue(h, 0);
  if (st == CLR_OK) st = clr_batch_get_results(h, loglike, logdet, quad, status);
  clr_batch_destroy(h);
  return st;
}

/* ======================================================================== */
/* CARMASolver (carma.h): host model + one-wave Kalman filter (carma.hip)    */
/* ======================================================================== */
struct clr_carma {
  clr::CarmaModel model;
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf dmodel, dt, dy, dyerr, dout;  // dout: [ll | status as int bits]
  bool model_resident = false;
};

clr_carma* clr_carma_create(double log_sigma, int p, const double* arparams, int q, const double* maparams,
                            int* status) {
  int st = CLR_OK;
  clr_carma* h = nullptr;
  if (p < 0 || q < 0 || (p > 0 && !arparams) || (q > 0 && !maparams)) {
    st = fail(CLR_INVALID_ARGUMENT, "bad CARMA parameter arrays");
  } else {
    h = new clr_carma();
    std::string err;
    st = clr::carma_setup(log_sigma, p, arparams, q, maparams, h->model, err);
    if (st != CLR_OK) {
      fail(st, err);
      delete h;
      h = nullptr;
    } else {
      h->device = g_device;
    }
  }
  if (status) *status = st;
  return h;
}

void clr_carma_destroy(clr_carma* h) {
  if (!h) return;
  if (h->stream || h->dmodel.p) {
    (void)hipSetDevice(h->device);
    for (DevBuf* b : {&h->dmodel, &h->dt, &h->dy, &h->dyerr, &h->dout}) b->release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
  }
  delete h;
}

int clr_carma_get_celerite_coeffs(const clr_carma* h, int* n_real, int* n_comp, double* a_real, double* c_real,
                                  double* a_comp, double* b_comp, double* c_comp, double* d_comp) {
  std::vector<double> v[6];
  clr::carma_celerite_coeffs(h->model, v);
  if (n_real) *n_real = (int)v[0].size();
  if (n_comp) *n_comp = (int)v[2].size();
  double* dst[6] = {a_real, c_real, a_comp, b_comp, c_comp, d_comp};
  for (int i = 0; i < 6; ++i)
    if (dst[i]) std::copy(v[i].begin(), v[i].end(), dst[i]);
  return CLR_OK;
}

int clr_carma_log_likelihood(clr_carma* h, int n_t, const double* t, int n_y, const double* y, int n_yerr,
                             const double* yerr, double* out) {
  if (n_y != n_t || n_yerr != n_t) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");  // carma.h:223
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->stream) HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const int p = h->model.p, n = n_t;
  if (n == 0 || p == 0) {
    // (an empty series: the filter loop does not run; p = 0 cannot happen, q < p)
    *out = -0.5 * (double)n * 1.8378770664093453;
    return CLR_OK;
  }
  if (!h->model_resident) {
    std::vector<double> pk;
    auto push = [&](const std::vector<std::complex<double>>& a) {
      for (const std::complex<double>& c : a) { pk.push_back(c.real()); pk.push_back(c.imag()); }
    };
    push(h->model.b); push(h->model.V); push(h->model.loglam);
    if ((st = upload(h->dmodel, pk.data(), pk.size(), h->stream)) != CLR_OK) return st;
    HIP_TRY(hipStreamSynchronize(h->stream));  // (pk is a local)
    h->model_resident = true;
  }
  if ((st = upload(h->dt, t, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = upload(h->dy, y, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = upload(h->dyerr, yerr, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = h->dout.reserve(2)) != CLR_OK) return st;
  clr::launch_carma_filter(n, p, h->dmodel.p, h->dt.p, h->dy.p, h->dyerr.p, h->dout.p,
                           reinterpret_cast<int*>(h->dout.p + 1), h->stream);
  HIP_TRY(hipGetLastError());
  double host[2] = {0.0, 0.0};
  HIP_TRY(hipMemcpyAsync(host, h->dout.p, sizeof(host), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int bad = 0;
  memcpy(&bad, &host[1], sizeof(int));
  if (bad) return fail(CLR_CARMA_INSTABILITY, "CARMA model encountered an instability");  // exceptions.h:8-12
  *out = host[0];
  return CLR_OK;
}

}  // extern "C"
