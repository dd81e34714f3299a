// celerite_amd/csrc/api_solver.hip -- C ABI of the object API (clr_solver_*): what the pybind11 module
// celerite_amd.solver binds in place of the reference's CholeskySolver<double> (celerite/solver.cpp:64-664).
#include "api_internal.h"

#include <mutex>
#include "clr_options.h"

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

// the end-state mismatch of the chunked replay that still counts as consistent (tuning: CLR_SOLVER_CERT_RESID)
static double solver_cert_resid() {
  if (const char* e = clr::option("CLR_SOLVER_CERT_RESID")) return atof(e);
  return 1e-11;
}

extern "C" {

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
                      &s->scratch, &s->scratch2, &s->scalars, &s->keep_diag, &s->keep_jitter, &s->ws_ends, &s->ws_elems, &s->ws_starts,
                      &s->ws_part, &s->ws_cond, &s->gradbuf, &s->gradws, &s->rhs, &s->ws_lvl_elems, &s->ws_lvl_starts})
      b->release();
    for (DevBuf& b : s->dot_buf) b.release();
    for (DevBuf& b : s->pred_buf) b.release();
    if (s->ws_flags) (void)hipFree(s->ws_flags);
    if (s->d_status) (void)hipFree(s->d_status);
    if (s->pin) (void)hipHostFree(s->pin);
    (void)hipStreamDestroy(s->stream);
  }
  delete s;
}


int clr_solver_compute(clr_solver* s, double jitter, int n_a_real, const double* a_real,
                       int n_c_real, const double* c_real, int n_a_comp, const double* a_comp,
                       int n_b_comp, const double* b_comp, int n_c_comp, const double* c_comp,
                       int n_d_comp, const double* d_comp, int n_A, const double* A, int U_rows,
                       int U_cols, const double* U, int V_rows, int V_cols, const double* V,
                       int n_x, const double* x, int n_diag, const double* diag) {
  const int N = n_x;
  s->computed = 0;  // cholesky.h:57
  s->refine_pending = 0;
  s->route_level = nullptr; s->route_nchunk = 0;
  s->big_maps_valid[0] = s->big_maps_valid[1] = false;
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
  if (J > CLR_MAX_WIDTH_ANY) return fail(CLR_UNSUPPORTED, "width above CLR_MAX_WIDTH_ANY");
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
  if (!has_general && clr::small_compute_supported(J_real, J_comp, N) && !clr::option("CLR_NO_SMALL_SOLVER")) {
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
    if ((st = stage_upload(s, s->keep_diag, diag, (size_t)N)) != CLR_OK) return st;   // (kept: the chunk-head pass reads them again)
    if ((st = stage_upload(s, s->keep_jitter, &jitter, 1)) != CLR_OK) return st;
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
    P.jitter = s->keep_jitter.p;
    P.a_real = g.a_real; P.c_real = g.c_real;
    P.a_comp = g.a_comp; P.b_comp = g.b_comp; P.c_comp = g.c_comp; P.d_comp = g.d_comp;
    // row-major arrays; with more than one chunk the kernels stage them through LDS
    if (use_rhs && (st = stage_upload(s, s->rhs, s->host_rhs.data(), (size_t)N)) != CLR_OK) return st;
    P.t = s->t.p; P.diag = s->keep_diag.p; P.y = use_rhs ? s->rhs.p : s->t.p;  // (without a hinted rhs y is irrelevant)
    if (P.nchunk > 1) {
      if ((st = s->ws_ends.reserve((size_t)P.nchunk * L->start_doubles)) != CLR_OK) return st;
      P.ends = s->ws_ends.p;
    }
    P.lane_is = 1; P.lane_cs = P.L;
    P.staged = P.nchunk > 1 ? 1 : 0;
    P.elems = s->ws_elems.p; P.starts = s->ws_starts.p; P.part = s->ws_part.p;
    P.flags = s->ws_flags;
    // the factor is wanted: always the exact replay, which overwrites the zero-start sums
    P.partx = P.part; P.flagsx = P.flags; P.need_exact = s->ws_flags + P.nchunk; P.force_exact = 1;
    P.out_ll = s->scalars.p; P.out_logdet = s->scalars.p + 1; P.out_quad = s->scalars.p + 2;
    P.out_status = reinterpret_cast<int*>(s->scalars.p + 3);
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    P.cond = s->ws_cond.p; P.cert_gamma = 1e7; P.cert_gamma_abs = 1e4; P.cert_eg = 3e-9; P.egerr = s->ws_cond.p + (size_t)P.nchunk * 3; P.cert_resid = solver_cert_resid(); P.logdet_only = use_rhs ? 0 : 1;
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
    s->route_level = P.need_exact; s->route_cond = P.cond; s->route_nchunk = P.nchunk;
    if (!h_status && P.ends) {  // the chunk heads once more, when the factor is first read (ensure_refined)
      s->refine_P = P;
      s->refine_P.y = s->t.p;       // (the pass writes factor entries only; the hinted rhs may be gone by then)
      s->refine_P.logdet_only = 1;
      s->refine_pending = 1;
    }
  } else if (!has_general && J >= 9 && J <= clr::wide_max_width()) {
    // widths 9..64 without general terms: the batched wide kernels on one problem -- one wave per
    // chunk with S distributed over the lanes, up to 16 chunks chained by the scan (widths <= 32),
    // the replay writing the factor in the reference's storage (instead of factor_generic_kernel:
    // one workgroup, five barriers per step)
    if ((st = stage_upload(s, s->keep_diag, diag, (size_t)N)) != CLR_OK) return st;   // (kept: the chunk-head pass reads them again)
    if ((st = stage_upload(s, s->keep_jitter, &jitter, 1)) != CLR_OK) return st;
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
      // (widths 33..64, round 5: the chunks are chained by wide_walk_kernel -- prefix + corrections, ~0.2 ms per chunk --
      //  and a step with riders costs ~2.2 us: a = 2.2 us, p = 200 us)
      nchunk = (int)lround(sqrt((double)N * (J <= 16 ? 0.096 : (J <= 32 ? 0.0208 : 0.011))));
      if (nchunk > N / 256) nchunk = N / 256;
      if (nchunk < 2 || N < 2048) nchunk = 1;  // (short series: the six launches of the chunked flow cost more)
      // round 4: the prefix is a parallel scan (wide_prefix_scan.hip: ceil(log2 nchunk) launches of 22 / 86 us instead of
      // nchunk steps of 14 / 50 us), so the chunks only have to amortise their own set-up -- 48 samples up to width 16, 96
      // above, at most 1024 / 512 chunks, at least 8 (profiles/r04v_single_wide_chunks.txt, r04z_single_wide_short.txt:
      // N = 1e5 width 16 3.0 -> 0.79 ms, width 32 6.9 -> 1.56 ms; N = 1000 0.74 -> 0.24 / 0.87 -> 0.58 ms)
      if (!clr::option("CLR_WIDE_PREFIX_WALK") && J <= 32) {
        const int cap = clr::wide_prefix_scan_max_chunks(J <= 16 ? 16 : 32), Lmin = J <= 16 ? 48 : 96;
        int nk = std::min(N / Lmin, cap);
        if (nk < 8 && N >= 8 * (J <= 16 ? 32 : 64)) nk = 8;
        if (nk >= 8) nchunk = nk;
      }
      if (const char* e = clr::option("CLR_SOLVER_WIDE_CHUNKS")) nchunk = std::max(1, std::min(atoi(e), N / 32));  // (tools/gpu_single_wide_chunks*.py)
    }
    P.L = (N + nchunk - 1) / nchunk;
    P.nchunk = (N + P.L - 1) / P.L;
    const size_t pc = (size_t)P.nchunk, JP = (size_t)clr::wide_padded_width(J), SZP = JP * (JP + 1) / 2;
    if ((st = s->ws_elems.reserve(pc * (JP * JP + JP + SZP + JP + SZP))) != CLR_OK) return st;
    if ((st = s->ws_starts.reserve(pc * (SZP + JP))) != CLR_OK) return st;
    const size_t scan_ws = clr::option("CLR_WIDE_PREFIX_WALK") ? 0 : clr::wide_prefix_scan_workspace(1, P.nchunk, (int)JP);  // the prefix as a parallel scan
    if (scan_ws && (st = s->ws_lvl_elems.reserve(scan_ws)) != CLR_OK) return st;
    P.scan_ws = scan_ws ? s->ws_lvl_elems.p : nullptr;
    if ((st = s->ws_part.reserve(pc * 4)) != CLR_OK) return st;
    if ((st = s->ws_cond.reserve(pc * 4)) != CLR_OK) return st;
    if ((st = reserve_flags(s->ws_flags, s->ws_flags_cap, 2 * pc + 1)) != CLR_OK) return st;
    const clr::GenericProblem g = generic_view(s);
    P.jitter = s->keep_jitter.p;
    P.a_real = g.a_real; P.c_real = g.c_real;
    P.a_comp = g.a_comp; P.b_comp = g.b_comp; P.c_comp = g.c_comp; P.d_comp = g.d_comp;
    if (use_rhs && (st = stage_upload(s, s->rhs, s->host_rhs.data(), (size_t)N)) != CLR_OK) return st;
    P.t = s->t.p; P.diag = s->keep_diag.p; P.y = use_rhs ? s->rhs.p : s->t.p;  // (without a hinted rhs y is irrelevant)
    if (P.nchunk > 1) {  // (refine_samples stays 0: wide_flow does not run the pass itself, ensure_refined does later)
      if ((st = s->ws_ends.reserve(2 * pc * (SZP + JP))) != CLR_OK) return st;
      P.ends = s->ws_ends.p;
      P.ends_alt = s->ws_ends.p + pc * (SZP + JP);  // (the output check's second buffer: BatchParams::head_check)
    }
    P.lane_is = 1; P.lane_cs = P.L;
    P.elems = s->ws_elems.p; P.starts = s->ws_starts.p;
    P.part = s->ws_part.p; P.partx = s->ws_part.p + pc * 2;
    P.flags = s->ws_flags; P.flagsx = s->ws_flags + pc; P.need_exact = s->ws_flags + 2 * pc;
    P.cond = s->ws_cond.p; P.cert_gamma = 1e7; P.cert_gamma_abs = 1e4; P.cert_eg = 3e-9; P.egerr = s->ws_cond.p + (size_t)P.nchunk * 3; P.cert_resid = solver_cert_resid(); P.logdet_only = use_rhs ? 0 : 1;
    P.force_exact = 1;       // the factor is wanted: every chunk is replayed (and checked against the scan)
    P.wide_materialize = 1;
    P.head_cap = clr::output_check_cap(); P.head_tol = clr::output_check_tol();  // (a state mismatch no output sees: BatchParams::head_check)
    P.coop_prefix = P.scan_ws ? 2 : 1;  // (2: the parallel prefix, wide_prefix_scan.hip)
    P.out_ll = s->scalars.p; P.out_logdet = s->scalars.p + 1; P.out_quad = s->scalars.p + 2;
    P.out_status = reinterpret_cast<int*>(s->scalars.p + 3);
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    HIP_TRY(hipMemsetAsync(P.need_exact, 0, sizeof(int), stream));  // (one chunk: no prefix kernel clears it)
    {
      const int wst = wide_flow(P, J_real, J_comp, stream, nullptr);
      if (wst != CLR_OK) return wst;
    }
    if (P.ends) {
      s->refine_P = P;
      s->refine_P.y = s->t.p;
      s->refine_P.logdet_only = 1;
      s->refine_pending = 2;  // (dropped below when the factorisation failed)
    }
    s->route_level = P.need_exact; s->route_cond = P.cond; s->route_nchunk = P.nchunk;
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
    bool rows_quad = false;
    // Above width 128 the row-distributed kernel's workgroups (4 .. 64 of them) spin on each other: they must all be
    // resident at once.  One such kernel always is (<= 64 workgroups on 256 compute units); several at a time -- solvers
    // of other host threads -- could each hold a part of the chip and wait for the rest.  They are serialised here, up to
    // the synchronisation below (the bounded spin turns what another PROCESS could still cause into CLR_HIP_ERROR).
    static std::mutex rows_many_mutex;
    std::unique_lock<std::mutex> rows_lock(rows_many_mutex, std::defer_lock);
    // (general terms at ANY total width: even at width 5 the padded width-128 step of the rows kernel, 1.6 us, is shorter
    //  than the LDS-resident kernel's five barriers, 1.9 .. 5.6 us at widths 5 .. 40 -- tools/gpu_rows_min_width.py)
    int rows_min = 1;
    if (const char* e = clr::option("CLR_ROWS_MIN_WIDTH")) rows_min = std::max(1, atoi(e));
    if (J >= rows_min && clr::factor_rows_supported(J) && !clr::option("CLR_NO_ROWS_KERNEL")) {
      if (J > 128) rows_lock.lock();
      // S in the registers of 1 .. 64 workgroups (rows_kernels.hip; round 6: width 128 20.5 -> ~1 us per step)
      if ((st = s->ws_elems.reserve(clr::factor_rows_workspace_doubles(J))) != CLR_OK) return st;
      double dmax = 0.0;
      for (int j = 0; j < J_comp; ++j) { const double m = fabs(d_comp[j]); if (!(m <= dmax)) dmax = m; }
      const int fast = (dmax * max_abs(x, N) < CLR_FAST_TRIG_LIMIT) ? 1 : 0;
      // (a hinted right-hand side: its quadratic form comes out of the same pass, as on the chunked routes)
      if (use_rhs && (st = stage_upload(s, s->rhs, s->host_rhs.data(), (size_t)N)) != CLR_OK) return st;
      clr::launch_factor_rows(g, fast, use_rhs ? s->rhs.p : nullptr, s->ws_elems.p, s->phi.p, s->u.p, s->W.p, s->D.p, s->d_status, s->scalars.p, stream);
      rows_quad = use_rhs;
    } else if (J > CLR_MAX_WIDTH) {  // S (J^2 doubles) in HBM / L2 instead of LDS (huge_kernels.hip)
      if ((st = s->ws_elems.reserve(clr::factor_huge_workspace_doubles(J))) != CLR_OK) return st;
      clr::launch_factor_huge(g, s->ws_elems.p, s->phi.p, s->u.p, s->W.p, s->D.p, s->d_status, s->scalars.p, stream);
    } else
    clr::launch_factor_generic(g, s->phi.p, s->u.p, s->W.p, s->D.p, s->d_status, s->scalars.p,
                               stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h_status, s->d_status, sizeof(int), hipMemcpyDeviceToHost, stream));
    double two[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(two, s->scalars.p, (rows_quad ? 2 : 1) * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    h_logdet = two[0];
    if (rows_quad && h_status == 0) { s->cached_quad = two[1]; s->have_quad = true; }
  }

  if (h_status == 3) return fail(CLR_HIP_ERROR, "the workgroups of the row-distributed factorisation lost each other (rows_kernels.hip)");
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
  if (JR + 2 * JC + JG > CLR_MAX_WIDTH_ANY) return fail(CLR_UNSUPPORTED, "grad_log_likelihood supports widths up to 1024");
  const int G = 1 + 2 * JR + 4 * JC;
  if (n_grad != G || !value || !grad) return fail(CLR_INVALID_ARGUMENT, "grad must hold 1 + 2 J_real + 4 J_comp values");
  if ((st = ensure_stream(s)) != CLR_OK) return st;
  hipStream_t stream = s->stream;

  const int Wc = JR + 2 * JC, Wt = Wc + JG;
  const bool narrow_plan = !has_general && JG == 0 && Wc >= 1 && Wc <= 8 && N >= 1024;
  // widths 9..32, and general terms up to a total width of 32: the wide scan + chunk-wise forward-mode tangents
  // (wide_batch_grad); chunk count for ONE problem from profiles/r04k_wide_grad_chunks.txt
  // (round 6) widths 33..64 without general terms: the same at the padded width 64, one direction per tangent wave
  const bool wide_plan = !narrow_plan && Wc >= 1 && (Wt <= 32 || (Wt <= 64 && JG == 0)) && (Wc >= 9 || JG > 0) && N >= 4096;
  if ((narrow_plan || wide_plan) && !clr::option("CLR_GRAD_SEQUENTIAL")) {
    // parallel in n: the scan + the chunk-wise tangents (clr_batch_grad) on a one-problem plan
    if (!s->grad_plan || s->grad_N != N || s->grad_JR != JR || s->grad_JC != JC || s->grad_wide != wide_plan) {
      if (s->grad_plan) clr_batch_destroy(s->grad_plan);
      s->grad_plan = clr_batch_create(1, N, JR, JC, s->device);
      s->grad_series.clear();
      s->grad_N = N; s->grad_JR = JR; s->grad_JC = JC; s->grad_wide = wide_plan;
      if (s->grad_plan && wide_plan) {
        // (the gradient's time is the tangent waves', one per (direction, chunk), and a tangent step costs the same in the
        //  first chunk as in any other: chunks of EQUAL length, not the evaluation's longer riderless first chunk)
        s->grad_plan->wide_first_ratio = 1.0;
        s->grad_plan->wide_first_ratio64 = 1.0;
        // chunks: the tangent pass dominates (one wave per (pair of partials, chunk), ~1.2 us per step whatever else the
        // SIMD holds): exactly one wave per SIMD -- ceil(G / 2) x nc <= 1024 -- as long as a chunk keeps >= 768 samples
        // (profiles/r04s_wide_grad_two_directions.txt: 17 x 60 = 1020 waves 4.04 ms, 17 x 64 = 1088 waves 4.72 ms -- the 64
        // SIMDs with a second wave finish last)
        int nc = std::min(1024 / ((G + 1) / 2), N / 768);
        // (widths 33..64: a wave per direction, S and dS of its row in 512 registers -- one wave per SIMD; the
        //  width-64 scan walks its chunks with one workgroup: a few dozen at most)
        if (Wt > 32) {
          // SEVERAL full rounds of the chip rather than one partly filled: with G x nc just below a multiple of 1024 the
          // tangent pass costs the same wave-steps, and the evaluation's scan -- one workgroup per chunk -- gets short chunks
          // (profiles/r06zz2_widegrad_kernel_trace_stats.txt: 7 chunks at width 64, G = 129: tangents 67 ms on 903 waves, the
          // scan 19 ms on 7 workgroups).  The largest count up to 32 whose last round is at least 93 % full.
          // (chunks of at least 2048 samples: below, the walk over the chunks and the riders outweigh the shorter scan --
          //  N = 2e4 at width 64: 21 ms with 7 chunks, 24 with 23)
          const int single = std::min(1024 / G, N / 768);  // (one round of waves)
          const int cap = std::min(32, N / 2048);
          nc = single;
          if (cap > single) {
            double best = 0.0;
            for (int c = cap; c >= 4; --c) {
              const long waves = (long)G * c;
              const double eff = (double)waves / (double)(((waves + 1023) / 1024) * 1024);
              if (eff >= 0.93) { nc = c; break; }
              if (eff > best) { best = eff; nc = c; }
            }
          }
        }
        nc = std::max(4, std::min(nc, 128));
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
  if (Wt > 64 || clr::option("CLR_GRAD_ANY_WIDTH")) {
    // above width 64 (round 6): one workgroup per direction, S and dS in an HBM / L2 workspace (grad_any_kernels.hip)
    if ((st = s->gradws.reserve(clr::grad_any_workspace_doubles(Wt, G))) != CLR_OK) return st;
    if (clr::launch_grad_any(P, s->gradws.p, stream) != 0) return fail(CLR_HIP_ERROR, "grad_log_likelihood: the any-width kernel could not be configured");
  } else
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
static bool big_sweep(const clr_solver* s) { return clr::bigsweep_supported(s->N, s->J) && !clr::option("CLR_NO_BIG_SWEEP"); }
static bool sweep_scan_ok(const clr_solver* s) {
  return clr::sweep_scan_supported(s->N, s->J) || clr::wsweep_scan_supported(s->N, s->J) || big_sweep(s);
}
// widths above 64: the affine scans of bigsweep_kernels.hip (`in` and `out` must be different arrays)
static int big_sweep_scan(clr_solver* s, int nrhs, const double* in, double* out, double* quad, int backward) {
  const int nc0 = clr::bigsweep_chunks(s->N, s->J);
  const int L = (s->N - 1 + nc0 - 1) / nc0, nchunk = (s->N - 1 + L - 1) / L;
  int st;
  if (!s->big_maps_valid[backward]) {
    if ((st = s->big_maps[backward].reserve(clr::bigsweep_maps_doubles(s->J, nchunk))) != CLR_OK) return st;
    clr::launch_bigsweep_maps(s->N, s->J, nchunk, L, backward, s->phi.p, s->u.p, s->W.p, s->big_maps[backward].p, s->stream);
    s->big_maps_valid[backward] = true;
  }
  const int SLICE = 4096;
  for (int r0 = 0; r0 < nrhs; r0 += SLICE) {
    const int nr = std::min(SLICE, nrhs - r0);
    clr::SweepParams P;
    memset(&P, 0, sizeof(P));
    P.N = s->N; P.J = s->J; P.nrhs = nr; P.nchunk = nchunk; P.L = L;
    P.phi = s->phi.p; P.u = s->u.p; P.W = s->W.p; P.D = s->D.p;
    P.in = in + (size_t)r0 * s->N;
    P.out = out ? out + (size_t)r0 * s->N : nullptr;
    P.quad = quad ? quad + r0 : nullptr;
    P.backward = backward;
    if ((st = s->ws_elems.reserve(clr::bigsweep_workspace_doubles(s->J, nchunk, nr))) != CLR_OK) return st;
    clr::launch_bigsweep_scan(P, s->big_maps[backward].p, s->ws_elems.p, s->stream);
  }
  return CLR_OK;
}
static int sweep_scan(clr_solver* s, int nrhs, const double* in, double* out, double* quad, int backward) {
  if (big_sweep(s)) return big_sweep_scan(s, nrhs, in, out, quad, backward);
  const bool wide = clr::wsweep_scan_supported(s->N, s->J);
  const int SLICE = 16384;  // right-hand sides per launch (grid.y / workspace bound); stream order keeps the slices apart
  for (int r0 = 0; r0 < nrhs; r0 += SLICE) {
    const int nr = std::min(SLICE, nrhs - r0);
    clr::SweepParams P;
    memset(&P, 0, sizeof(P));
    P.N = s->N; P.J = s->J; P.nrhs = nr;
    P.nchunk = wide ? clr::wsweep_chunks(s->N, s->J) : clr::sweep_chunks(s->N);
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

static int ensure_refined(clr_solver* s);

int clr_solver_debug_route(const clr_solver* cs, int* level, int* nchunk, double* residual) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  if (level) *level = -1;
  if (nchunk) *nchunk = 0;
  if (residual) *residual = 0.0;
  if (!s->computed || !s->route_level || s->route_nchunk < 2) return CLR_OK;
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  int lv = -1;
  std::vector<double> rec((size_t)s->route_nchunk * 3);
  HIP_TRY(hipMemcpyAsync(&lv, s->route_level, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  if (s->route_cond) HIP_TRY(hipMemcpyAsync(rec.data(), s->route_cond, rec.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  double r = 0.0;
  for (int c = 0; c < s->route_nchunk; ++c) { const double v = rec[(size_t)c * 3 + 2]; r = (v != v) ? INFINITY : std::max(r, v); }
  if (level) *level = lv;
  if (nchunk) *nchunk = s->route_nchunk;
  if (residual) *residual = r;
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
  if ((st = ensure_refined(s)) != CLR_OK) return st;
  if ((st = upload(s->scratch, b, (size_t)s->N, s->stream)) != CLR_OK) return st;
  if ((st = s->scalars.reserve(8)) != CLR_OK) return st;
  if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, 1, s->scratch.p, nullptr, s->scalars.p, 0)) != CLR_OK) return st;
  } else if (s->J > CLR_MAX_WIDTH) {
    clr::launch_dot_solve_huge(s->N, s->J, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p, s->scalars.p, s->stream);
  } else {
    clr::launch_dot_solve(s->N, s->J, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                          s->scalars.p, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, s->scalars.p, sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return CLR_OK;
}

// the factor is about to be read: the pending pass over the chunk heads (clr_solver::refine_pending) runs first -- every
// chunk once more from the state the previous chunk's replay reached (fixup_steps = the chunk length)
static int ensure_refined(clr_solver* s) {
  if (!s->refine_pending || !s->computed) return CLR_OK;
  clr::BatchParams F = s->refine_P;
  F.fixup_steps = F.L + (F.L0 > F.L ? F.L0 - F.L : 0);
  if (s->refine_pending == 1) {
    const clr::BatchLaunchers* L = clr::find_batch_launchers(s->J_real, s->J_comp);
    if (L) L->replay(F, 1, s->stream);
  } else {
    clr::launch_wide_loglike(F, s->J_real, s->J_comp, s->stream);
  }
  s->refine_pending = 0;
  HIP_TRY(hipGetLastError());
  return CLR_OK;
}

static int sweep_common(clr_solver* s, int rows, int nrhs, const double* in) {
  if (rows != s->N) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");
  if (!s->computed) return fail(CLR_NOT_COMPUTED, "you must call 'compute' first");
  int st = ensure_stream(s);
  if (st != CLR_OK) return st;
  if ((st = ensure_refined(s)) != CLR_OK) return st;
  const size_t n = (size_t)s->N * (size_t)std::max(nrhs, 0);
  if ((st = upload(s->scratch, in, n, s->stream)) != CLR_OK) return st;
  return s->scratch2.reserve(n);
}

int clr_solver_solve(const clr_solver* cs, int b_rows, int nrhs, const double* b, double* x) {
  clr_solver* s = const_cast<clr_solver*>(cs);
  int st = sweep_common(s, b_rows, nrhs, b);
  if (st != CLR_OK) return st;
  if (nrhs <= 0) return CLR_OK;
  double* result = s->scratch2.p;
  if (big_sweep(s)) {  // (its passes do not work in place: b -> scratch2 -> scratch)
    if ((st = sweep_scan(s, nrhs, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;   // :240-248
    if ((st = sweep_scan(s, nrhs, s->scratch2.p, s->scratch.p, nullptr, 1)) != CLR_OK) return st;   // :249-259
    result = s->scratch.p;
  } else if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, nrhs, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;   // :240-248
    if ((st = sweep_scan(s, nrhs, s->scratch2.p, s->scratch2.p, nullptr, 1)) != CLR_OK) return st;  // :249-259
  } else if (s->J > CLR_MAX_WIDTH) {
    clr::launch_solve_huge(s->N, s->J, nrhs, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p, s->scratch2.p, s->stream);
  } else {
    clr::launch_solve(s->N, s->J, nrhs, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p,
                      s->scratch2.p, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(x, result, sizeof(double) * (size_t)s->N * nrhs,
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
  if (J > CLR_MAX_WIDTH_ANY) return fail(CLR_UNSUPPORTED, "width above CLR_MAX_WIDTH_ANY");
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

  // this call must not disturb a previously computed factor: buffers of its own -- kept in the solver object between
  // calls (round 4: eleven hipMalloc / hipFree pairs per call were ~0.7 of the 1.1-1.2 ms of a call at N = 1e5)
  DevBuf &coeffs = s->dot_buf[0], &tt = s->dot_buf[1], &dU = s->dot_buf[2], &dV = s->dot_buf[3], &phi = s->dot_buf[4],
         &u = s->dot_buf[5], &v = s->dot_buf[6], &dg = s->dot_buf[7], &zin = s->dot_buf[8], &yout = s->dot_buf[9],
         &ws = s->dot_buf[10];
  auto cleanup = [&]() {};  // (released with the solver)
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
  if (big_sweep(s)) {  // (not in place: y -> scratch2 -> scratch, then back to scratch2 where predict reads alpha)
    if ((st = sweep_scan(s, 1, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;
    if ((st = sweep_scan(s, 1, s->scratch2.p, s->scratch.p, nullptr, 1)) != CLR_OK) return st;
    HIP_TRY(hipMemcpyAsync(s->scratch2.p, s->scratch.p, sizeof(double) * (size_t)s->N, hipMemcpyDeviceToDevice, stream));
  } else if (sweep_scan_ok(s)) {
    if ((st = sweep_scan(s, 1, s->scratch.p, s->scratch2.p, nullptr, 0)) != CLR_OK) return st;
    if ((st = sweep_scan(s, 1, s->scratch2.p, s->scratch2.p, nullptr, 1)) != CLR_OK) return st;
  } else if (s->J > CLR_MAX_WIDTH) {
    clr::launch_solve_huge(s->N, s->J, 1, s->phi.p, s->u.p, s->W.p, s->D.p, s->scratch.p, s->scratch2.p, stream);
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
    // chunks of 16 samples (at most 8192): the prefix over the chunks is a parallel scan (predict_prefix_kernel), the
    // prediction points and the chunk summaries walk <= one chunk each (profiles/r04z_predict_chunks.txt: N = 1e5, M = 2e4,
    // width 8 2.1 -> 0.84 ms, width 32 4.9 -> 1.5 ms; rounds 2-3: 1.8 sqrt(N) chunks, their prefix one thread's walk)
    pchunk = std::max(1, std::min(s->N / 16, 8192));
    if (const char* e = clr::option("CLR_PREDICT_CHUNKS")) pchunk = std::max(1, std::min(atoi(e), s->N / 8));  // (tools/gpu_predict_chunks.py)
    pL = (s->N + pchunk - 1) / pchunk;
    pchunk = (s->N + pL - 1) / pL;
    if ((st = s->ws_elems.reserve(clr::predict_workspace_doubles(pchunk, s->J_real + 2 * s->J_comp))) != CLR_OK) return st;
  }
  DevBuf &dxs = s->pred_buf[0], &dpred = s->pred_buf[1];  // (kept between calls)
  if ((st = upload(dxs, xs, (size_t)M, stream)) != CLR_OK) return st;
  if ((st = dpred.reserve((size_t)M)) != CLR_OK) return st;
  hipError_t e = hipMemsetAsync(dpred.p, 0, sizeof(double) * (size_t)M, stream);
  const clr::GenericProblem g = generic_view(s);
  if (scan) clr::launch_predict_scan(g, s->scratch2.p, M, dxs.p, dpred.p, s->ws_elems.p, pchunk, pL, stream);
  else clr::launch_predict(g, s->scratch2.p, M, dxs.p, dpred.p, stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(pred, dpred.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
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
  if ((st = ensure_refined(s)) != CLR_OK) return st;
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
  s->big_maps_valid[0] = s->big_maps_valid[1] = false;
  s->refine_pending = 0;
  s->N = N;
  s->J = J;
  s->log_det = log_det;
  s->J_real = s->J_comp = s->J_general = 0;
  if (!computed) return CLR_OK;
  if (J < 0 || J > CLR_MAX_WIDTH_ANY || N < 1) return fail(CLR_INVALID_ARGUMENT, "Invalid state!");
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

}  // extern "C"
