// celerite_amd/csrc/api_grad.hip -- C ABI of the batched gradient (clr_batch_grad, clr_batch_grad_log_likelihood and
// their settings): reverse / forward mode on the narrow plans (clr_grad_kernels.h), the chunk-parallel forward mode
// at widths 9..32 and with general terms (wide_grad_kernels.hip), the sequential tangent kernel as the fallback.
#include "api_internal.h"
#include "clr_options.h"

extern "C" {

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
  if (!general && Wt <= 32 && (h->launch || h->nchunk < 2))
    return fail(CLR_UNSUPPORTED, "the plan gradient at widths 9..32 needs a chunked plan: use clr_batch_grad_log_likelihood");
  // widths 33..64: chunked plans take the chunk-wise tangents at the padded width 64 (round 6: the riders' elimination
  // in 133 KB of LDS under a workgroup of 256 threads, wide_grad_riders64_kernel); a plan of one chunk runs the SEQUENTIAL
  // tangent kernel on the plan's resident series and coefficients (one wave per problem and direction,
  // solver.cpp:347-463 as it stands).  CLR_GRAD_SEQUENTIAL: the sequential kernel for every problem (cross-checks).
  const bool seq_all = Wt > 32 && (h->nchunk < 2 || clr::option("CLR_GRAD_SEQUENTIAL") != nullptr);
  if (seq_all && general)
    return fail(CLR_UNSUPPORTED, "the plan gradient with general terms covers total widths up to 32");
  // (the tangent kernels are built up to the padded width 64: rows beyond it would be dropped while the diagonal sums
  //  still covered every term -- a wrong gradient with CLR_OK)
  if (Wt > clr::wide_max_width())
    return fail(CLR_UNSUPPORTED, "the plan gradient covers widths up to 64");
  // The evaluation behind the gradient must hand out FINAL values: no level-1 problem may be left pending for a side
  // plan (defer_runs) -- the walk reads ll / status straight from the device, before any resolve could run
  int st = CLR_OK;
  if (!seq_all) {
    h->grad_scan_only = true;
    st = clr_batch_enqueue(h, 0);
    h->grad_scan_only = false;
  }
  if (st != CLR_OK) return st;
  clr::BatchParams P0, P;
  if ((st = batch_params(h, 0, P0)) != CLR_OK) return st;
  if (general) general_wide_params(h, P0, P); else P = P0;
  const size_t B = (size_t)h->B, NG = 1 + 2 * (size_t)h->J_real + 4 * (size_t)h->J_comp;
  const int JP = Wt <= 16 ? 16 : (Wt <= 32 ? 32 : 64);
  const size_t pc = B * (size_t)P.nchunk, RID = 2 * (size_t)JP * JP + JP, OUT = (size_t)JP * JP + JP + 2;
  if (!seq_all && (st = h->g_riders.reserve(pc * RID)) != CLR_OK) return st;
  if (!seq_all && (st = h->g_out.reserve(pc * NG * OUT)) != CLR_OK) return st;
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
  G.only_level = seq_all ? nullptr : P.need_exact;
  G.nchunk = P.nchunk; G.L = P.L; G.L0 = P.L0; G.JP = JP;
  G.starts = P.starts; G.rec = h->g_out.p;
  G.out_value = d_value; G.out_grad = d_grad; G.out_status = d_status;

  if (!seq_all) {
    if (clr::launch_wide_grad_riders(W, h->stream) != 0) return fail(CLR_HIP_ERROR, "the riders kernel could not be configured (LDS)");
    clr::launch_grad_chunked(G, h->stream);
    if (clr::launch_wide_grad_walk(W, h->stream) != 0) return fail(CLR_HIP_ERROR, "the walk kernel could not be configured (LDS)");
  }
  clr::launch_grad(G, h->stream);  // (sequential form: only the problems with level >= 2 -- or, widths 33..64, all)
  HIP_TRY(hipGetLastError());
  std::vector<double> back(B * (NG + 2));
  HIP_TRY(hipMemcpyAsync(back.data(), h->g_res.p, back.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int* hst = reinterpret_cast<const int*>(back.data() + B + B * NG);
  int nfb = 0;
  std::vector<int> levels(B, 2);
  if (!seq_all) HIP_TRY(hipMemcpy(levels.data(), P.need_exact, B * sizeof(int), hipMemcpyDeviceToHost));
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
      const double walk = rev ? 0.2 : 0.5 + 4.5 * w2 * J / 8.0;  // per chunk: a wave per problem / a thread per (problem, direction)
      // (reverse, >= 256 gradient chunks: the adjoint walk in two levels -- 2 seg + ng / seg dependent steps, seg = sqrt(ng / 2),
      //  two more launches; profiles/r04z_single_grad_trace.txt)
      const double walk_total = (rev && ng >= 256) ? 0.3 * 2.0 * std::sqrt(2.0 * ng) + 12.0 : ng * walk;
      const double tm = std::max(1.0, waves / 1024.0) * k * h->L * step + walk_total;
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
    P.g_span = 1;
    if (h->grad_K > 0) {
      P.g_K = (int)std::min<long>(h->grad_K, Lg);
      nalloc = (Lg + P.g_K - 1) / P.g_K;
    } else {
      P.g_K = 0;
      // states the rule asks for fewer than g_span steps after a stored one are rebuilt forwards by the sweep
      // (GradStore::span; profiles/r04t_grad_rebuild_span.txt); CLR_GRAD_REBUILD_SPAN: the tools' A/B knob
      P.g_span = h->grad_rebuild_span;
      if (const char* e = clr::option("CLR_GRAD_REBUILD_SPAN")) P.g_span = std::max(1, std::min(atoi(e), 200));
      if ((st = grad_chunk_spans(h)) != CLR_OK) return st;
      double need = 0.0;
      for (size_t b = 0; b < B; ++b) {
        const double v = h->host_cmax[b] * h->grad_span[h->t_stride == 0 ? 0 : b] * P.g_m / CLR_GRAD_GROWTH_BUDGET;
        if (!(v <= need)) need = v;
      }
      nalloc = (need == need && need < (double)Lg) ? (long)(2.0 * std::ceil(need)) + 8 : Lg;
      nalloc = std::min<long>(nalloc, Lg);
      if (P.g_span > 1) nalloc = std::min<long>(nalloc, Lg / P.g_span + 2);
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
    // one long series: the adjoint walk over its thousands of gradient chunks in two levels (clr_grad_kernels.h) --
    // groups of sqrt(chunks / 2): compose + walk the groups + fan out = 2 seg + chunks / seg dependent steps
    P.g_seg = P.g_nchunk >= 256 ? std::max(16, (int)std::sqrt(0.5 * P.g_nchunk)) : 0;
    const size_t ngr = P.g_seg ? (size_t)((P.g_nchunk + P.g_seg - 1) / P.g_seg) : 0;
    const size_t nslab = (size_t)((P.g_nchunk + 255) / 256);
    const size_t small = pc * (RID + 3 * (SZ + J) + NG + 2) + B + B * ngr * (RID + SZ + J) + B * nslab * 33;
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
      P.g_grp_riders = ngr ? P.g_drift_max + B : nullptr;
      P.g_grp_adj = ngr ? P.g_grp_riders + B * ngr * RID : nullptr;
      P.g_slab = P.g_drift_max + B + B * ngr * (RID + SZ + J);
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
  if (J_real + 2 * J_comp <= 8 && N >= 512 && !clr::option("CLR_GRAD_SEQUENTIAL")) {
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

}  // extern "C"
