// celerite_amd/csrc/api_batch.hip -- C ABI of the batched plans (clr_batch_*): HBM residency, path selection
// (scan pipeline, warm-started recurrence, one-launch path, wide kernels, general terms), evaluation and results.
#include "api_internal.h"
#include "clr_options.h"
#include "clr_group_hooks.h"

extern "C" {

static void mp_release(clr_batch* h);

/* ---- batched log-likelihood ---------------------------------------------------- */
clr_batch* clr_batch_create(int B, int N, int J_real, int J_comp, int device) {
  if (B < 1 || N < 1 || J_real < 0 || J_comp < 0) {
    fail(CLR_INVALID_ARGUMENT, "clr_batch_create: bad sizes");
    return nullptr;
  }
  const clr::BatchLaunchers* L = clr::find_batch_launchers(J_real, J_comp);
  const int width = J_real + 2 * J_comp;
  if (!L && (width < 1 || width > CLR_MAX_WIDTH)) {
    fail(CLR_UNSUPPORTED, "batched path supports widths 1..128 (J_real + 2 J_comp)");
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
  mp_release(h);
  if (h->rescue) clr_batch_destroy(h->rescue);
  h->rescue = nullptr;
  if (h->rescue_idx) (void)hipFree(h->rescue_idx);
  for (DevBuf* b : {&h->coeffs, &h->t, &h->diag, &h->y, &h->tT, &h->dT, &h->yT,
                    &h->elems, &h->starts, &h->part, &h->partx, &h->cond, &h->out, &h->phi, &h->u, &h->W, &h->D,
                    &h->fphi, &h->fu, &h->fW, &h->fD, &h->lvl_elems, &h->lvl_starts, &h->wstarts, &h->wends,
                    &h->wpart, &h->wresid, &h->wT, &h->wD, &h->wY, &h->gA, &h->gU, &h->gV, &h->g_riders, &h->g_out,
                    &h->g_res, &h->g_rec, &h->g_ck, &h->bs_rm, &h->bs_x, &h->bs_M, &h->bs_off, &h->bs_starts, &h->bs_decay, &h->bs_y, &h->ends, &h->sT, &h->sD, &h->sY})
    b->release();
  if (h->flags) (void)hipFree(h->flags);
  if (h->wints) (void)hipFree(h->wints);
  if (h->g_ckflag) (void)hipFree(h->g_ckflag);
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  for (hipEvent_t& e : h->bs_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (h->pin) (void)hipHostFree(h->pin);
  clr::staging_destroy(h->staging);
  h->scan.release();
  for (DevBuf* b : {&h->gen_elems, &h->gen_starts, &h->gen_part, &h->gen_cond, &h->gen_scan}) b->release();
  if (h->gen_flags) (void)hipFree(h->gen_flags);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int warm_plan_chunks(clr_batch* h);
static int warm_resolve(clr_batch* h, bool* pin_current);
static int warm_scan_spans(clr_batch* h);
static void warm_select(clr_batch* h);
static int plan_general_chunks(clr_batch* h);

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
    else if (h->J > 32) {
      // widths 33..64 (round 5): the summarize keeps S and A^T in 256 registers per lane -- ONE wave per SIMD -- so a round
      // of the chip is B x nchunk = 1024 waves; the chunks are chained by a walk per problem (~0.2 ms per chunk:
      // wide64_kernels.hip), so at most 16, each of >= 1024 samples.  Above 512 problems two chunks no longer pay.
      if (nchunk <= 0) {
        nchunk = h->B <= 512 ? std::max(2, 1024 / h->B) : 1;
        if (nchunk > 16) nchunk = 16;
        while (nchunk > 1 && h->N / nchunk < 1024) --nchunk;
      }
    } else if (nchunk <= 0) {
      // One round of two waves per SIMD: B x nchunk = 2048 waves.  (Round 2 used 4096 / B -- two rounds of half the
      //  length -- because the checked replay of borderline problems, 7 ms for config 4, got shorter with the chunks;
      //  with the round-3 routing that family is settled from the chunk summaries and the sequential prefix + the
      //  correct phase decide: B = 256: 8 chunks 17.1 ms, 16 chunks 18.5, 10 chunks 22.8 (2560 waves = a second,
      //  nearly empty round), B = 512: 4 chunks 32.2 ms, 8 chunks 32.7; profiles/r03_wide_chunks.txt)
      nchunk = h->B <= 1024 ? 2048 / h->B : 1;  // (above 1024 problems one sweep per problem already fills a round)
      if (nchunk < 2) nchunk = 1;
      if (nchunk > 16) nchunk = 16;
      while (nchunk > 1 && h->N / nchunk < 512) --nchunk;
      // few problems: the prefix is a parallel scan (wide_prefix_scan.hip) as long as B x nchunk <= 1024 / 512 -- chunks of
      // >= 64 / 96 samples instead of 16 long ones (one series of 1e5 samples: 1024 / 512 chunks)
      const int cap = clr::wide_prefix_scan_max_chunks(h->J <= 16 ? 16 : 32), Lmin = h->J <= 16 ? 64 : 96;  // (r04z_single_wide_short.txt)
      if (h->coop_prefix == 2 && h->B < 32 && std::min(cap / h->B, h->N / Lmin) > nchunk) nchunk = std::min(cap / h->B, h->N / Lmin);
      // 32..256 problems (profiles/r04zz_wide_midbatch.txt, N = 1e5): 16 long chunks leave the chip half empty -- width <= 16:
      // 64 chunks at B = 32 (4.1 -> 1.6 ms), 32 at B = 64..256 (3.9 -> 2.8, 5.2 -> 4.9, 9.9 -> 8.5 ms); width 32: 32 chunks at
      // B = 32 (5.4 -> 3.5 ms), the old counts above (the walk's 50 us per chunk and problem)
      // (width <= 16 up to 1024 problems: B = 512 19.6 -> 15.8 ms with 32 chunks, B = 1024 36.2 -> 30.6 ms with 16)
      if (h->coop_prefix == 2 && h->B >= 32 && h->B <= (h->J <= 16 ? 1024 : 256)) {
        int want = nchunk;
        if (h->J <= 16) want = h->B <= 512 ? std::max(32, 2048 / h->B) : 16;
        else if (h->B == 32) want = 32;
        if (want > nchunk && h->N / want >= 512) nchunk = want;
      }
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
  if (const char* e = clr::option("CLR_WIDE_FIRST_RATIO")) h->wide_first_ratio = atof(e);  // (tuning runs only)
  if (const char* e = clr::option("CLR_WIDE_FIRST_RATIO64")) h->wide_first_ratio64 = atof(e);
  const double first_ratio = h->J > 32 ? h->wide_first_ratio64 : h->wide_first_ratio;
  if (!h->launch && h->nchunk > 1 && first_ratio > 1.0) {
    // wide scan: the first chunk's summarize carries no riders (wide_scan_body, RIDERS == false) and costs
    // ~1 / wide_first_ratio of a later chunk's per sample: it gets that many more samples, so that all waves of the
    // one round finish together.  Chunks 1.. have exactly L samples, the first one the rest.
    const int nc = h->nchunk;
    int L = (int)ceil(h->N / (nc - 1 + first_ratio));
    L = (L + 7) & ~7;
    const long first = (long)h->N - (long)(nc - 1) * L;
    if (L >= 64 && first >= L) { h->L = L; h->L0 = (int)first; }
  }
  h->relayout_pending = true;
  h->grad_span_valid = false;
  h->have_factor = false;  // its layout depends on the chunking
  h->factor_valid = false;
  const size_t pc = (size_t)h->B * h->nchunk;
  h->plan = clr::plan_prefix(h->nchunk, 0, 0);
  h->scan_ws_doubles = 0;
  if (h->launch) {
    if ((st = h->elems.reserve(pc * h->launch->elem_doubles)) != CLR_OK) return st;
    if ((st = h->starts.reserve(pc * h->launch->start_doubles)) != CLR_OK) return st;
    // The prefix plan fixes the ORDER in which chunk elements are composed, i.e. the bits of every result: it must not
    // depend on how many problems happen to share the plan (a slice of a sharded batch, a side plan of pending problems
    // and the unsharded plan must agree bit for bit).  Its time model is asked for the batch size the automatic chunking
    // pairs with this chunk count -- about 65536 (problem, chunk) lanes, one round of the chip -- instead of the plan's own.
    h->plan = clr::plan_prefix(h->nchunk, h->plan_levels, h->plan_g, std::max(1, 65536 / std::max(1, h->nchunk)), h->J);
    size_t le = 0, ls = 0;
    clr::multilevel_workspace(h->plan, h->J, &le, &ls);
    if (le && (st = h->lvl_elems.reserve((size_t)h->B * le)) != CLR_OK) return st;
    if (ls && (st = h->lvl_starts.reserve((size_t)h->B * ls)) != CLR_OK) return st;
  } else if (h->nchunk > 1) {  // elements / start states at the padded width (16 or 32)
    const size_t JP = (size_t)clr::wide_padded_width(h->J), SZ = JP * (JP + 1) / 2;
    if ((st = h->elems.reserve(pc * (JP * JP + JP + SZ + JP + SZ))) != CLR_OK) return st;
    if ((st = h->starts.reserve(pc * (SZ + JP))) != CLR_OK) return st;
    // few problems with many chunks: the prefix as a parallel scan (wide_prefix_scan.hip); its level buffers
    h->scan_ws_doubles = clr::wide_prefix_scan_workspace(h->B, h->nchunk, (int)JP);
    if (h->scan_ws_doubles && (st = h->lvl_elems.reserve(h->scan_ws_doubles)) != CLR_OK) return st;
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
  if (h->J_general > 0 && (st = plan_general_chunks(h)) != CLR_OK) return st;  // (its chunking follows the plan's settings)
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
  h->small_copy_pending = true;
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
  h->warm_eligible = 0;
  h->warm_eligible_total = -1;
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
  // (a batch with only a few eligible problems is not worth a second set of launches; a slice of a larger batch waits
  //  for the count over the whole batch: clr_group::set_warm_eligible_total)
  h->warm_eligible = (long)eligible;
  h->warm_eligible_total = -1;
  h->warm_active = h->group_B > 0 ? false : eligible * 2 >= B;
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
  // from here on the old series is being overwritten: the plan has NO series until every copy and scan below has
  // succeeded (a failed upload must not leave have_series set over half-written arrays), and nothing derived from
  // the old one -- interleaved copies, warm-up spans and their selection -- survives
  h->have_series = false;
  h->factor_inputs_changed = true;
  h->warm_active = false;
  h->warm_span.clear();
  h->relayout_pending = true;
  h->warm_copy_pending = true;
  h->small_copy_pending = true;
  h->grad_span_valid = false;
  const clr::CopyJob jobs[3] = {{h->t.p, t, count(t_stride)}, {h->diag.p, diag, count(diag_stride)}, {h->y.p, y, count(y_stride)}};
  const size_t total = (jobs[0].n + jobs[1].n + jobs[2].n) * sizeof(double);
  bool staged = false;
  if (total >= ((size_t)32 << 20)) {
    // large series: NT host threads stage pieces through pinned buffers, their DMAs share the link (clr_series_io.h);
    // no pinned memory to be had (64 MB): the plain copies below
    if (clr::staging_create(h->staging, h->device) == 0) {
      const int e = clr::upload_parallel(h->staging, jobs, 3);
      if (e != 0) return fail(CLR_HIP_ERROR, hipGetErrorString((hipError_t)e));
      staged = true;
    } else {
      clr::staging_destroy(h->staging);
      (void)hipGetLastError();
    }
  }
  if (!staged) {
    for (const clr::CopyJob& j : jobs)
      if (j.n) HIP_TRY(hipMemcpyAsync(j.dst, j.src, j.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  h->t_stride = t_stride;
  h->diag_stride = diag_stride;
  h->y_stride = y_stride;
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
    // the smallest step over the finite differences (the device's fmin skips NaN); a negative one means "not sorted"
    // whatever else the series holds -- the reference's np.any(np.diff(t) < 0), celerite.py:126-129 -- and only a
    // series without a negative step reports its NaN times as NaN
    const double finite_min = N > 1 ? dmin : 0.0;
    h->dtmin = (finite_min < 0.0) ? finite_min : (nan ? NAN : finite_min);
  }
  h->have_series = true;
  if ((st = warm_scan_spans(h)) != CLR_OK) { h->have_series = false; return st; }
  h->set_series_host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  h->grad_span_valid = false;
  h->relayout_pending = true;
  h->warm_copy_pending = true;
  h->small_copy_pending = true;
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
  h->pin_results = false;
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
  h->factor_inputs_changed = true;
  h->pin_results = false;  // (the staging buffer is about to carry the coefficients)
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

int clr_batch_set_exact(clr_batch* h, int force) {
  h->force_exact = force ? 1 : 0;
  return CLR_OK;
}

int clr_batch_get_exact_count(clr_batch* h, int* count) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;  // (pending problems are settled first: their routes too)
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
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;  // (pending problems are settled first: their routes too)
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
  if (h->J_general > 0) return plan_general_chunks(h);  // (the parallel prefix asks for other chunk counts)
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
    *kind = (h->nchunk > 1 && (h->summarize_mode < 0 || h->summarize_mode == 2) &&
             (h->J_general > 0 ? lazy_eligible(h) : lazy_eligible_wide(h))) ? 2 : 0;
  return CLR_OK;
}

// A plan with general terms on the wide kernels (total width <= 64): its own chunking and workspace, derived from the
// plan's CURRENT settings -- batch size, explicit chunk count, prefix mode, first-chunk ratio.  Called by
// clr_batch_set_general and again by every setter that changes one of those (clr_batch_set_chunks and, through it,
// clr_batch_set_prefix_plan; clr_batch_set_prefix_mode), so that the chunking does not depend on the order of the calls.
static int plan_general_chunks(clr_batch* h) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  h->gen_nchunk = 0;
  const int J_general = h->J_general;
  if (J_general <= 0) return CLR_OK;
  // Total widths up to 64 run on the wave-per-(problem, chunk) kernels (wide_kernels.hip, GEN flavour): the general rows
  // are one more row class there.  Chunks as for any wide plan: one round of two waves per SIMD, the scan up to width 32.
  const int Wt = h->J + J_general;
  if (Wt <= clr::wide_max_width()) {
    int nchunk = (Wt <= clr::wide_scan_max_width() && h->B <= 1024) ? 2048 / h->B : 1;
    if (Wt > 32) nchunk = h->B <= 512 ? std::max(2, 1024 / h->B) : 1;  // (one wave per SIMD at total widths 33..64: clr_batch_set_chunks)
    if (nchunk > 16) nchunk = 16;
    while (nchunk > 1 && h->N / nchunk < (Wt > 32 ? 1024 : 512)) --nchunk;
    const int cap = clr::wide_prefix_scan_max_chunks(Wt <= 16 ? 16 : 32), Lmin = Wt <= 16 ? 64 : 96;
    if (Wt <= 32 && h->coop_prefix == 2 && h->B < 32 && std::min(cap / h->B, h->N / Lmin) > nchunk)
      nchunk = std::min(cap / h->B, h->N / Lmin);  // (few problems: the parallel prefix, clr_batch_set_chunks)
    if (h->warm_explicit_chunks > 0 && Wt <= clr::wide_scan_max_width())  // (an explicit clr_batch_set_chunks is honoured, whenever it was made)
      nchunk = std::min(h->warm_explicit_chunks, std::max(1, h->N / 64));
    if (nchunk < 1) nchunk = 1;
    int L = (h->N + nchunk - 1) / nchunk;
    if (nchunk > 1) L = (L + 7) & ~7;
    nchunk = (h->N + L - 1) / L;
    int L0 = 0;
    const double first_ratio = Wt > 32 ? h->wide_first_ratio64 : h->wide_first_ratio;
    if (nchunk > 1 && first_ratio > 1.0) {  // (the riderless, longer first chunk: clr_batch_set_chunks)
      int L2 = (int)ceil(h->N / (nchunk - 1 + first_ratio));
      L2 = (L2 + 7) & ~7;
      const long first = (long)h->N - (long)(nchunk - 1) * L2;
      if (L2 >= 64 && first >= L2) { L = L2; L0 = (int)first; }
    }
    const size_t pc = (size_t)h->B * nchunk, JP = (size_t)clr::wide_padded_width(Wt), SZ = JP * (JP + 1) / 2;
    if (nchunk > 1) {
      if ((st = h->gen_elems.reserve(pc * (JP * JP + JP + SZ + JP + SZ))) != CLR_OK) return st;
      if ((st = h->gen_starts.reserve(pc * (SZ + JP))) != CLR_OK) return st;
      h->gen_scan_ws_doubles = clr::wide_prefix_scan_workspace(h->B, nchunk, (int)JP);
      if (h->gen_scan_ws_doubles && (st = h->gen_scan.reserve(h->gen_scan_ws_doubles)) != CLR_OK) return st;
    } else {
      h->gen_scan_ws_doubles = 0;
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
  if ((st = plan_general_chunks(h)) != CLR_OK) { h->J_general = 0; return st; }
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
  return !h->pipeline_pinned && sel_B(h) <= (h->J >= 3 ? 256 : 1024);
}

// The one-launch path's view of the series: lane = chunk of ceil(N / 256) samples, so the row-major arrays would be read
// with a stride of one chunk between the lanes -- every load instruction touches 64 cache lines of which it uses 8 bytes
// each, and the 768 lines a workgroup walks (256 lanes x 3 arrays) do not fit the CU's vector L1: each line is fetched
// from L2 sixteen times (BASELINE configs[1]: 3.8 MB per problem through a 64-B/clk port -- the summarize phase was
// bound by exactly that, profiles/r06l_config1_ab.txt).  The kernel therefore reads a chunk-interleaved copy
// [problem][i][chunk] made once per set_series (launch_relayout with the path's own chunking, padded as the plan's copy):
// one coalesced 512-B load per array, wave and step.
static int small_params(clr_batch* h, clr::BatchParams& Sp) {
  h->in_fallback = true;  // (the plan's row-major arrays, no role split)
  int st = batch_params(h, 0, Sp);
  h->in_fallback = false;
  if (st != CLR_OK) return st;
  const int T = 256, L = (h->N + T - 1) / T;
  const long cells = (long)L * T;
  auto nsrc = [&](long sd) { return (size_t)(sd == 0 ? 1 : h->B); };
  if ((st = h->sT.reserve(nsrc(h->t_stride) * cells)) != CLR_OK) return st;
  if ((st = h->sD.reserve(nsrc(h->diag_stride) * cells)) != CLR_OK) return st;
  if ((st = h->sY.reserve(nsrc(h->y_stride) * cells)) != CLR_OK) return st;
  if (h->small_copy_pending) {
    struct { DevBuf* src; DevBuf* dst; long stride; int pad; } jobs[3] = {
        {&h->t, &h->sT, h->t_stride, 1}, {&h->diag, &h->sD, h->diag_stride, 2}, {&h->y, &h->sY, h->y_stride, 0}};
    for (auto& j : jobs)
      clr::launch_relayout(j.src->p, j.stride, j.dst->p, j.stride ? cells : 0, j.stride ? h->B : 1, h->N, L, T, j.pad, h->stream);
    h->small_copy_pending = false;
  }
  Sp.t = h->sT.p; Sp.diag = h->sD.p; Sp.y = h->sY.p;
  Sp.t_stride = h->t_stride ? cells : 0;
  Sp.diag_stride = h->diag_stride ? cells : 0;
  Sp.y_stride = h->y_stride ? cells : 0;
  Sp.lane_is = T; Sp.lane_cs = 1;
  return CLR_OK;
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


// ---- materialising runs as a pipeline over groups of problems ---------------------------------------------------
// The materialising step is a fp64-VALU-bound pass (summarize: the chunk elements) followed by an HBM-bound one
// (replay: 8 N (3 J + 1) bytes of factor per problem).  Back to back on one stream they add up -- 2.2 + 3.8 ms at the
// headline shape, 48 % of the HBM roofline for the whole step.  Here the batch is cut into G contiguous groups of
// problems; the summarize of group g + 1 runs while group g is replayed, on streams that own DISJOINT sets of compute
// units (hipExtStreamCreateWithCUMask): the role-split summarize fills a CU completely (2 x 256 registers per SIMD,
// 160 KB of LDS), so without the masks concurrency only serialises whole kernels.  The prefix and the corrections of
// a group (small, latency-bound) run on a third, unmasked stream between the two.
static void mp_release(clr_batch* h) {
  for (hipStream_t s : h->mp_s) if (s) (void)hipStreamDestroy(s);
  h->mp_s.clear();
  if (h->mp_p) (void)hipStreamDestroy(h->mp_p);
  if (h->mp_r) (void)hipStreamDestroy(h->mp_r);
  h->mp_p = h->mp_r = nullptr;
  for (hipEvent_t e : h->mp_ev) (void)hipEventDestroy(e);
  h->mp_ev.clear();
}

// CU `i` of the mask belongs to the summarize set iff ((i / 8) + (i % 8)) % 16 < k: whichever way the runtime numbers
// the mask bits over the 8 XCDs (round-robin or XCD-major) both sets are spread over every XCD -- each XCD's L2 and
// fabric port then carry their share of the replay's stores.
static void mp_masks(int total_cus, int summarize_cus, std::vector<uint32_t>& ms, std::vector<uint32_t>& mr) {
  const int words = (total_cus + 31) / 32, k = summarize_cus / 16;
  ms.assign(words, 0u);
  mr.assign(words, 0u);
  for (int i = 0; i < total_cus; ++i) {
    const bool sset = ((i / 8) + (i % 8)) % 16 < k;
    (sset ? ms : mr)[i / 32] |= 1u << (i % 32);
  }
}

static int mp_prepare(clr_batch* h) {
  const int G = h->mp_groups;
  if (!h->mp_s.empty() && (int)h->mp_ev.size() == 2 * G + 2) return CLR_OK;
  mp_release(h);
  int ncu = 0;
  HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));
  std::vector<uint32_t> ms, mr;
  const bool masked = h->mp_cus >= 16 && h->mp_cus <= ncu - 16;
  if (masked) mp_masks(ncu, h->mp_cus, ms, mr);
  h->mp_s.assign((size_t)h->mp_nstreams, nullptr);
  for (hipStream_t& s : h->mp_s) {
    if (masked) HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)ms.size(), ms.data()));
    else HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  if (masked) HIP_TRY(hipExtStreamCreateWithCUMask(&h->mp_r, (uint32_t)mr.size(), mr.data()));
  else HIP_TRY(hipStreamCreateWithFlags(&h->mp_r, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->mp_p, hipStreamNonBlocking));
  h->mp_ev.assign((size_t)2 * G + 2, nullptr);
  for (hipEvent_t& e : h->mp_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return CLR_OK;
}

// the kernels' view of problems [b0, b0 + Bg) of the plan: every per-problem pointer advanced, B = Bg (the levels of the
// multi-level prefix are laid out per view: [Bg][n_l] blocks back to back inside the group's own region)
static clr::BatchParams group_view(const clr_batch* h, const clr::BatchParams& P, int b0, int Bg) {
  clr::BatchParams G = P;
  const size_t o = (size_t)b0, JR = (size_t)h->J_real, JC = (size_t)h->J_comp, J = (size_t)h->J, nc = (size_t)P.nchunk;
  const size_t E = (size_t)h->launch->elem_doubles, S = (size_t)h->launch->start_doubles, cells = (size_t)P.L * nc;
  G.B = Bg;
  G.jitter += o; G.a_real += o * JR; G.c_real += o * JR;
  G.a_comp += o * JC; G.b_comp += o * JC; G.c_comp += o * JC; G.d_comp += o * JC;
  G.t += (long)o * P.t_stride; G.diag += (long)o * P.diag_stride; G.y += (long)o * P.y_stride;
  G.elems += o * nc * E; G.starts += o * nc * S;
  G.part += o * nc * 2; G.partx += o * nc * 2;
  if (G.cond) G.cond += o * nc * 3;
  if (G.egerr) G.egerr += o * nc;
  G.flags += o * nc; G.flagsx += o * nc; G.need_exact += o;
  size_t le = 0, ls = 0;
  clr::multilevel_workspace(P.plan, (int)J, &le, &ls);
  if (G.lvl_elems) G.lvl_elems += o * le;
  if (G.lvl_starts) G.lvl_starts += o * ls;
  // (the lean factor layout stores W and D only: phi / u are never reserved there)
  if (G.phi) G.phi += o * J * cells;
  if (G.u) G.u += o * J * cells;
  if (G.W) G.W += o * J * cells;
  if (G.D) G.D += o * cells;
  if (G.ends) G.ends += o * nc * S;
  G.out_ll += o; G.out_logdet += o; G.out_quad += o; G.out_status += o;
  G.only_pending = 0;
  return G;
}

// replay mode of a materialising run: 2 the four arrays chunk-interleaved, 3 the lean layout (W, D only)
static int replay_mode(const clr_batch* h, int materialize) {
  return materialize ? (h->factor_layout == 1 ? 3 : 2) : 0;
}

// the fix-up pass of a materialising run: the first `factor_refine` samples of every chunk from the previous chunk's
// replayed end state (BatchParams::ends / fixup_steps, replay_kernel)
static void refine_chunk_heads(const clr_batch* h, clr::BatchParams R, int materialize, hipStream_t s) {
  R.fixup_steps = R.refine_samples;
  h->launch->replay(R, replay_mode(h, materialize), s);
}

static bool mp_runs(const clr_batch* h, int materialize) {
  return materialize && h->launch && h->mp_groups >= 2 && h->nchunk > 1 && h->B >= h->mp_groups && h->J_general == 0;
}

static int materialize_pipeline(clr_batch* h, const clr::BatchParams& P) {
  int st = mp_prepare(h);
  if (st != CLR_OK) return st;
  const int G = h->mp_groups;
  const clr::BatchParams R = replay_view(h, P, 1);
  HIP_TRY(hipEventRecord(h->mp_ev[0], h->stream));  // (whatever the plan's stream holds -- uploads, the relayout -- comes first)
  for (hipStream_t s : h->mp_s) HIP_TRY(hipStreamWaitEvent(s, h->mp_ev[0], 0));
  HIP_TRY(hipStreamWaitEvent(h->mp_p, h->mp_ev[0], 0));
  HIP_TRY(hipStreamWaitEvent(h->mp_r, h->mp_ev[0], 0));
  for (int g = 0; g < G; ++g) {
    int lo = 0, hi = 0;
    clr_shard_bounds(h->B, G, g, &lo, &hi);
    const clr::BatchParams Pg = group_view(h, P, lo, hi - lo), Rg = group_view(h, R, lo, hi - lo);
    hipStream_t ss = h->mp_s[(size_t)g % h->mp_s.size()];
    h->launch->summarize(Pg, ss);
    HIP_TRY(hipEventRecord(h->mp_ev[1 + 2 * g], ss));
    HIP_TRY(hipStreamWaitEvent(h->mp_p, h->mp_ev[1 + 2 * g], 0));
    h->launch->prefix(Pg, h->mp_p);
    h->launch->correct(Pg, h->mp_p);
    HIP_TRY(hipEventRecord(h->mp_ev[2 + 2 * g], h->mp_p));
    HIP_TRY(hipStreamWaitEvent(h->mp_r, h->mp_ev[2 + 2 * g], 0));
    h->launch->replay(Rg, replay_mode(h, 1), h->mp_r);
    if (Rg.ends) refine_chunk_heads(h, Rg, 1, h->mp_r);
    h->launch->sequential(Pg, replay_mode(h, 1), h->mp_r);
  }
  HIP_TRY(hipEventRecord(h->mp_ev[2 * G + 1], h->mp_r));  // (the replay stream's last group closes every chain of events)
  HIP_TRY(hipStreamWaitEvent(h->stream, h->mp_ev[2 * G + 1], 0));
  h->factor_is_lean = h->factor_layout == 1;
  h->factor_inputs_changed = false;
  h->factor_valid = true;
  h->bs_M_valid = false;
  clr::launch_finalize(P, h->stream);
  HIP_TRY(hipGetLastError());
  return CLR_OK;
}

int clr_batch_set_rescue(clr_batch* h, int mode) {
  if (mode < -1 || mode > 1) return fail(CLR_INVALID_ARGUMENT, "rescue mode: -1 (auto), 0 (inline chunked replay) or 1 (whenever possible)");
  int st = warm_resolve(h, nullptr);
  if (st != CLR_OK) return st;
  h->rescue_mode = mode;
  return CLR_OK;
}

int clr_batch_get_rescue(const clr_batch* h, int* last_count, long* total, int* nchunk, int* chunk_len) {
  if (last_count) *last_count = h->rescue_last;
  if (total) *total = h->rescue_total;
  if (nchunk) *nchunk = h->rescue ? h->rescue->nchunk : 0;
  if (chunk_len) *chunk_len = h->rescue ? h->rescue->L : 0;
  return CLR_OK;
}

int clr_batch_set_factor_layout(clr_batch* h, int layout) {
  if (layout != 0 && layout != 1) return fail(CLR_INVALID_ARGUMENT, "factor layout: 0 (phi, u, W, D) or 1 (lean: W, D)");
  if (layout == 1 && !h->launch) return fail(CLR_UNSUPPORTED, "the lean factor layout covers widths 1..8 (wider plans write the reference's storage)");
  if (layout != h->factor_layout) { h->have_factor = false; h->factor_valid = false; }
  h->factor_layout = layout;
  return CLR_OK;
}

int clr_batch_set_factor_refine(clr_batch* h, int samples) {
  if (samples < 0) return fail(CLR_INVALID_ARGUMENT, "factor refine: samples per chunk head >= 0");
  h->factor_refine = samples;
  return CLR_OK;
}

int clr_batch_get_factor_bytes(const clr_batch* h, size_t* bytes_per_problem) {
  if (!bytes_per_problem) return fail(CLR_INVALID_ARGUMENT, "bytes_per_problem is null");
  const size_t cells = h->launch ? (size_t)h->L * h->nchunk : (size_t)h->N, J = (size_t)h->J;
  *bytes_per_problem = 8 * cells * ((h->launch && h->factor_layout == 1) ? (J + 1) : (3 * J + 1));
  return CLR_OK;
}

int clr_batch_set_materialize_pipeline(clr_batch* h, int groups, int summarize_cus, int summarize_streams) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (groups < 0 || groups == 1 || groups > 64 || summarize_cus < 0 || summarize_streams < 1 || summarize_streams > 8)
    return fail(CLR_INVALID_ARGUMENT, "materialize pipeline: groups 0 (off) or 2..64, summarize_cus >= 0, 1..8 summarize streams");
  if (!h->launch && groups) return fail(CLR_UNSUPPORTED, "the materialising pipeline covers widths 1..8");
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (hipStream_t s : h->mp_s) if (s) HIP_TRY(hipStreamSynchronize(s));
  mp_release(h);
  h->mp_groups = groups;
  h->mp_cus = (summarize_cus / 16) * 16;
  h->mp_nstreams = summarize_streams;
  return CLR_OK;
}

int clr_batch_debug_cu_census(clr_batch* h, int which, int* cus_per_xcc /* [8] */) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!cus_per_xcc || which < 0 || which > 2) return fail(CLR_INVALID_ARGUMENT, "cu census: which = 0 (plan), 1 (summarize), 2 (replay)");
  hipStream_t s = h->stream;
  if (which > 0) {
    if (h->mp_groups < 2) return fail(CLR_INVALID_ARGUMENT, "cu census: no materialising pipeline is set");
    if ((st = mp_prepare(h)) != CLR_OK) return st;
    s = which == 1 ? h->mp_s[0] : h->mp_r;
  }
  int* seen = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&seen), 8 * 256 * sizeof(int)));
  HIP_TRY(hipMemsetAsync(seen, 0, 8 * 256 * sizeof(int), s));
  clr::launch_cu_census(seen, 8192, 20000, s);
  std::vector<int> host(8 * 256);
  HIP_TRY(hipMemcpyAsync(host.data(), seen, host.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(seen);
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    for (int c = 0; c < 256; ++c) n += host[(size_t)x * 256 + c] > 0;
    cus_per_xcc[x] = n;
  }
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
  if (!P.defer_level1) h->rescue_last = 0;  // (nothing is re-planned behind this evaluation)
  h->pin_results = false;
  const bool all_marks = h->prof_on != 2;
  auto mark = [&](int i) { if (ev && (all_marks || i == 1 || i == 2)) (void)hipEventRecord(ev[i], h->stream); };
  h->evaluated = true;
  if (h->J_general > 0 || h->J > clr::wide_max_width()) {
    // general terms -- and (round 5) celerite-only kernels of widths 65..128, which have no wave-per-problem kernel --: the
    // any-width sequential recurrence, one workgroup per problem with S in LDS (generic_loglike_batch_kernel)
    if (materialize) return fail(CLR_UNSUPPORTED, "materialising runs with general terms or above width 64: use CholeskySolver");
    h->warm_inflight = false;
    h->small_inflight = false;
    if (h->gen_nchunk > 0 && h->general_route != 1) {
      // the wide kernels with the general rows as a third row class (chunked scan up to total width 32)
      clr::BatchParams W;
      general_wide_params(h, P, W);
      mark(0);
      if ((st = wide_flow(W, h->J_real, h->J_comp, h->stream, ev)) != CLR_OK) return st;
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
    // (round 6) total widths 33 .. 128: S in the registers of the problem's workgroup (rows_kernels.hip: 1.6 us per sample at
    // width 128 where the LDS-resident kernel takes 20)
    if (clr::factor_rows_batch_supported(h->J + h->J_general) && !clr::option("CLR_NO_ROWS_KERNEL"))
      clr::launch_factor_rows_batch(G, P.fast_trig, h->stream);
    else
    clr::launch_generic_loglike_batch(G, h->stream);
    mark(2); mark(3); mark(4); mark(5); mark(6);
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  if (!h->launch) {
    mark(0);
    if ((st = wide_launch(h, P, ev)) != CLR_OK) return st;
    if (materialize) { h->factor_is_lean = false; h->factor_inputs_changed = false; h->factor_valid = true; h->bs_M_valid = false; }
    h->rescue_inflight = P.defer_level1 != 0;
    HIP_TRY(hipGetLastError());
    return CLR_OK;
  }
  mark(0);
  h->warm_inflight = false;
  h->small_inflight = false;
  h->rescue_inflight = false;
  if (!warm_runs(h, materialize) && small_runs(h, materialize)) {
    clr::BatchParams Sp;
    if ((st = small_params(h, Sp)) != CLR_OK) return st;
    mark(1);
    clr::launch_small_batch(h->J_real, h->J_comp, Sp, 256, h->stream);
    mark(2); mark(3); mark(4); mark(5); mark(6);
    h->small_inflight = true;  // (pending problems are settled like the warm path's: warm_resolve)
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
  if (mp_runs(h, materialize)) {  // groups of problems: summarize of one beside the replay of the previous (above)
    mark(1); mark(2); mark(3); mark(4);
    if ((st = materialize_pipeline(h, P)) != CLR_OK) return st;
    mark(5); mark(6);
    return CLR_OK;
  }
  mark(1);
  h->launch->summarize(P, h->stream);
  mark(2);
  h->launch->prefix(P, h->stream);
  mark(3);
  h->launch->correct(P, h->stream);  // (also on forced-exact runs: flags + conditioning record)
  mark(4);
  h->launch->replay(replay_view(h, P, materialize), replay_mode(h, materialize), h->stream);  // forced-exact / materialising runs only
  if (P.ends) refine_chunk_heads(h, replay_view(h, P, materialize), materialize, h->stream);
  h->launch->sequential(P, replay_mode(h, materialize), h->stream);  // flagged / ill-conditioned problems only
  if (materialize) { h->factor_is_lean = h->factor_layout == 1; h->factor_inputs_changed = false; h->factor_valid = true; h->bs_M_valid = false; }
  mark(5);
  clr::launch_finalize(P, h->stream);
  mark(6);
  h->rescue_inflight = P.defer_level1 != 0;
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

// ---- level-1 problems re-planned as a small plan of their own ----------------------------------
// The evaluation left the problems its conditioning record sends to the checked chunked replay PENDING (defer_runs,
// finalize_kernel).  Their checked replay is three passes -- summarize, prefix, replay + end-state check -- over the
// chunks of a plan sized for THEM: n problems cut into ~1024 / n chunks each, so that the chip is full and a chunk is a
// tenth of the parent's.  Same routes, same certificates (a side plan runs forced-exact: every chunk replayed from its
// scanned start state and checked, mismatches go to its own sequential recurrence), cholesky.h:176 semantics included.
static int rescue_inline(clr_batch* h) {
  // too many pending problems for a side plan to pay: the inline chunked replay after all (the flow's own tail)
  clr::BatchParams P;
  int st = batch_params(h, 0, P);
  if (st != CLR_OK) return st;
  P.defer_level1 = 0;
  if (!h->launch) {
    clr::launch_wide_loglike(P, h->J_real, h->J_comp, h->stream);
    clr::launch_wide_check_replay(P, h->stream);
    clr::launch_finalize(P, h->stream);
    clr::BatchParams S = P;
    S.nchunk = 1; S.L = P.N; S.L0 = 0; S.seq_only = 1; S.force_exact = 1;
    clr::launch_wide_loglike(S, h->J_real, h->J_comp, h->stream);
  } else {
    h->launch->replay(replay_view(h, P, 0), 0, h->stream);
    h->launch->sequential(P, 0, h->stream);
    clr::launch_finalize(P, h->stream);
  }
  HIP_TRY(hipGetLastError());
  return CLR_OK;
}

// `n_total`: the pending problems of the WHOLE batch (the plan's own count unless it is a slice of a larger batch):
// side plan or inline replay, and the side plan's chunk count, follow that number -- never the slice's own -- so that
// a problem is re-planned the same way under any sharding.
static int rescue_run(clr_batch* h, const std::vector<int>& idx, long n_total) {
  const int n = (int)idx.size();
  h->rescue_total += n;
  if (n_total > std::max(1, sel_B(h) / 4) || n_total > 256) {
    h->rescue_last = -n;
    return rescue_inline(h);
  }
  h->rescue_last = n;
  int st;
  if (!h->rescue || h->rescue->B != n || h->rescue_plan_key != n_total) {
    if (h->rescue) clr_batch_destroy(h->rescue);
    h->rescue_plan_key = -1;
    h->rescue = clr_batch_create(n, h->N, h->J_real, h->J_comp, h->device);
    if (!h->rescue) return CLR_HIP_ERROR;
    clr_batch* r = h->rescue;
    r->is_rescue_plan = true;
    r->force_exact = 1;
    r->warm_mode = 0;
    r->small_mode = 0;
    int nchunk = 0;
    const int nt = (int)n_total;
    if (!h->launch) {  // wide: B x nchunk <= the parallel prefix's cap (1024 workgroups per level at width 32, 2048 below)
      const int JP = clr::wide_padded_width(h->J);
      if (JP <= 32) {
        nchunk = std::min(clr::wide_prefix_scan_cap(JP) / nt, clr::wide_prefix_scan_max_chunks(JP));
        nchunk = std::min(nchunk, h->N / (JP == 16 ? 64 : 96));
        if (nchunk < 8) nchunk = 0;  // (short series: the plan's own choice)
      }  // (widths 33..64: the plan's own rule -- 1024 / n chunks, at most 16, chained by the walk)
      if (nchunk == 0) {  // the automatic rule, asked as the side plan of the whole batch would ask it
        clr_batch* probe = (nt == n) ? nullptr : clr_batch_create(nt, h->N, h->J_real, h->J_comp, h->device);
        if (probe) { nchunk = probe->nchunk; clr_batch_destroy(probe); }
      }
    } else {
      nchunk = auto_chunks(nt, h->N, h->J, true);
    }
    // (a failed set-up must not leave a half-initialised side plan behind: the next resolve would take it for ready)
    if ((st = clr_batch_set_chunks(r, nchunk)) != CLR_OK) {
      clr_batch_destroy(h->rescue);
      h->rescue = nullptr;
      return st;
    }
    h->rescue_plan_key = n_total;
  }
  clr_batch* r = h->rescue;
  // the parent's settings that decide routes and kernels
  r->cert_resid = h->cert_resid; r->cert_gamma = h->cert_gamma; r->cert_gamma_abs = h->cert_gamma_abs; r->cert_eg = h->cert_eg;
  r->force_library_trig = h->force_library_trig;
  r->floor_tmax = sel_max(h->tmax, h->floor_tmax); r->floor_dxmax = sel_max(h->dxmax, h->floor_dxmax);
  r->floor_dmax = sel_max(h->dmax, h->floor_dmax); r->floor_cmax = sel_max(h->cmax, h->floor_cmax);
  r->tmax = h->tmax; r->dxmax = h->dxmax; r->dmax = h->dmax; r->cmax = h->cmax; r->dtmin = h->dtmin;
  if ((size_t)n > h->rescue_idx_cap) {
    if (h->rescue_idx) (void)hipFree(h->rescue_idx);
    h->rescue_idx = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->rescue_idx), (size_t)n * sizeof(int)));
    h->rescue_idx_cap = (size_t)n;
  }
  HIP_TRY(hipStreamSynchronize(r->stream));
  HIP_TRY(hipMemcpyAsync(h->rescue_idx, idx.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, r->stream));
  HIP_TRY(hipStreamSynchronize(r->stream));  // (pageable source)
  // series and coefficients of the n problems, device to device (the parent's stream is idle: warm_resolve synchronised it)
  const size_t N = (size_t)h->N;
  struct { DevBuf* src; DevBuf* dst; long stride; long* sub_stride; } arrs[3] = {
      {&h->t, &r->t, h->t_stride, &r->t_stride}, {&h->diag, &r->diag, h->diag_stride, &r->diag_stride}, {&h->y, &r->y, h->y_stride, &r->y_stride}};
  for (auto& a : arrs) {
    if (a.stride == 0) {  // one series shared by all problems: shared by the side plan's too
      if ((st = a.dst->reserve(N)) != CLR_OK) return st;
      HIP_TRY(hipMemcpyAsync(a.dst->p, a.src->p, N * sizeof(double), hipMemcpyDeviceToDevice, r->stream));
      *a.sub_stride = 0;
    } else {
      if ((st = a.dst->reserve((size_t)n * N)) != CLR_OK) return st;
      clr::launch_gather_series(a.src->p, a.stride, a.dst->p, h->rescue_idx, n, h->N, r->stream);
      *a.sub_stride = (long)N;
    }
  }
  r->have_series = true;
  r->relayout_pending = true;
  r->warm_copy_pending = true;
  r->small_copy_pending = true;
  r->grad_span_valid = false;
  r->factor_inputs_changed = true;
  const size_t total = (size_t)n * (2 * h->J_real + 4 * h->J_comp + 1);
  if ((st = r->coeffs.reserve(total)) != CLR_OK) return st;
  clr::launch_gather_coeffs(h->coeffs.p, r->coeffs.p, h->rescue_idx, h->B, n, h->J_real, h->J_comp, r->stream);
  r->host_cmin.assign((size_t)n, 0.0);
  r->host_cmax.assign((size_t)n, h->cmax);
  r->host_jitter.assign((size_t)n, 0.0);
  r->have_coeffs = true;
  if ((st = clr_batch_enqueue(r, 0)) != CLR_OK) return st;
  const size_t pc = (size_t)n * r->nchunk, pcp = (size_t)h->B * h->nchunk;
  clr::launch_scatter_results(r->out.p, r->flags + 2 * pc, n, h->out.p, h->flags + 2 * pcp, h->B, h->rescue_idx, r->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(r->stream));
  return CLR_OK;
}

// Problems the warm path could not settle (boundary mismatch, flagged pivot, not eligible) carry a pending status
// until the scan pipeline has run for them.  That pipeline reads the plan's CURRENT coefficients, series and
// chunking, so it must run before any of them changes: every state-changing entry point, clr_batch_synchronize and
// clr_batch_get_results call this first.  *pin_current (may be null): the pinned staging buffer holds the final
// results (ll | logdet | quad | status) of this evaluation.
//
// Two halves (a plan that is a slice of a larger batch has the sharded layer add up the counts between them,
// clr_group_hooks.h): resolve_begin waits for the evaluation and counts, resolve_finish acts on the counts of the
// whole batch.
static int resolve_begin(clr_batch* h, long* pending_out, long* eligible_out) {
  if (pending_out) *pending_out = 0;
  if (eligible_out) *eligible_out = 0;
  if (h->res_open) {  // (begin twice without a finish: the same counts again)
    if (pending_out) *pending_out = h->res_pending;
    if (eligible_out) *eligible_out = h->res_was_warm ? h->warm_eligible : 0;
    return CLR_OK;
  }
  if (!h->warm_inflight && !h->small_inflight && !h->rescue_inflight) return CLR_OK;
  // (the one-launch path of short narrow problems leaves pending problems the same way, but its outcome says nothing
  //  about the warm-up lengths: the warm path's statistics and its adaptation are not touched on its behalf)
  h->res_was_warm = h->warm_inflight;
  h->res_was_rescue = h->rescue_inflight;
  h->warm_inflight = false;
  h->small_inflight = false;
  h->rescue_inflight = false;
  const size_t B = (size_t)h->B, words = 3 * B + (B + 1) / 2;
  int st;
  if ((st = reserve_pinned(h, words)) != CLR_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int* stw = reinterpret_cast<const int*>(h->pin + 3 * B);
  long pending = 0;
  for (size_t b = 0; b < B; ++b) pending += stw[b] == clr::CLR_PENDING_STATUS;
  if (h->res_was_warm) {
    h->warm_fallbacks = (int)pending;
    h->warm_settled = (int)B - (int)pending;
  }
  h->res_pending = pending;
  h->res_open = true;
  h->pin_results = true;  // (final unless something is pending: resolve_finish re-reads then)
  if (pending_out) *pending_out = pending;
  if (eligible_out) *eligible_out = h->res_was_warm ? h->warm_eligible : 0;
  return CLR_OK;
}

// `pending_total`, `eligible_total`: the counts over the whole batch (the plan's own when it is the whole batch)
static int resolve_finish(clr_batch* h, long pending_total, long eligible_total) {
  if (!h->res_open) return CLR_OK;
  h->res_open = false;
  const size_t B = (size_t)h->B, words = 3 * B + (B + 1) / 2;
  const bool was_warm = h->res_was_warm, was_rescue = h->res_was_rescue;
  const long pending = h->res_pending;
  const int* stw = reinterpret_cast<const int*>(h->pin + 3 * B);
  int st;
  if (was_rescue) {
    // the scan pipeline itself ran: what is pending are its level-1 problems, re-planned with short chunks
    h->rescue_last = 0;
    if (pending) {
      std::vector<int> idx;
      for (size_t b = 0; b < B; ++b) if (stw[b] == clr::CLR_PENDING_STATUS) idx.push_back((int)b);
      h->pin_results = false;
      if ((st = rescue_run(h, idx, pending_total)) != CLR_OK) return st;
      HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      h->pin_results = true;
    }
    return CLR_OK;
  }
  if (pending) {
    h->pin_results = false;
    if ((st = warm_fallback(h)) != CLR_OK) return st;
    HIP_TRY(hipMemcpyAsync(h->pin, h->out.p, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->pin_results = true;
  }
  if (pending_total) {
    // many mismatches among the problems that did warm up: longer warm-ups from the next coefficients on
    const long failed = pending_total - ((long)sel_B(h) - eligible_total);
    if (was_warm && h->warm_mode < 0 && failed * 10 > eligible_total && h->warm_boost < clr_batch::WARM_NK - 1) ++h->warm_boost;
  } else if (was_warm && h->warm_mode < 0 && h->warm_boost > 0 && ++h->warm_clean >= 8) {
    --h->warm_boost;  // eight clean evaluations in a row: try the shorter warm-ups again
    h->warm_clean = 0;
  }
  if (pending_total && was_warm) h->warm_clean = 0;
  return CLR_OK;
}

static int warm_resolve(clr_batch* h, bool* pin_current) {
  if (pin_current) *pin_current = false;
  if (!h->res_open && !h->warm_inflight && !h->small_inflight && !h->rescue_inflight) {
    // (nothing in flight; an evaluation the sharded layer has already resolved left its results in the staging buffer)
    if (pin_current) *pin_current = h->pin_results;
    return CLR_OK;
  }
  long pending = 0, eligible = 0;
  int st = resolve_begin(h, &pending, &eligible);
  if (st != CLR_OK) return st;
  // (a slice of a larger batch resolved on its own -- clr_batch_run_timed's steps, a direct call -- acts on its own
  //  counts scaled to nothing: the slice's numbers stand for the batch's)
  if ((st = resolve_finish(h, pending, eligible)) != CLR_OK) return st;
  if (pin_current) *pin_current = h->pin_results;
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
  if (h->factor_is_lean && h->factor_inputs_changed)
    return fail(CLR_NOT_COMPUTED, "the lean factor's phi and u are regenerated from the plan's series and coefficients, "
                                  "which were replaced after the materialising run: materialise again");
  if (h->factor_is_lean) {
    // the lean layout holds W and D; phi and u are regenerated from the plan's times and the coefficients in force --
    // which must still be the ones of the materialising run (any change drops the factor: have_factor)
    clr::BatchParams P;
    if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
    h->launch->expand(P, p, h->t.p + (size_t)p * (size_t)h->t_stride, h->fphi.p, h->fu.p, h->fW.p, h->fD.p, h->stream);
  } else
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

// clr_batch_solve on a wide plan (widths 9..64): the factor lies in the reference's storage, problem after problem
// (wide_scan_kernel, MODE 0), and the two sweeps of CholeskySolver::solve are the object API's chunked affine scans
// (wsweep_kernels.hip: one wave per chunk, one lane per column of the chunk's map) launched ONCE for the whole batch --
// grid.z = problem (SweepParams::batch).  Chunks per problem: about two rounds of the chip's SIMDs over the batch
// (2048 / B, at most the single solver's 1024, at least 2).
static int wide_batch_solve(clr_batch* h, int nrhs, const double* b, double* x) {
  int st;
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  if (!h->have_factor || !h->factor_valid) return fail(CLR_NOT_COMPUTED, "no materialising run has been made (clr_batch_enqueue(h, 1))");
  if (h->J_general > 0 || h->J > clr::wide_max_width()) return fail(CLR_UNSUPPORTED, "clr_batch_solve covers celerite-only plans of widths 1..64");
  if (!clr::wsweep_scan_supported(h->N, h->J)) return fail(CLR_UNSUPPORTED, "clr_batch_solve on a wide plan needs N >= 512 (shorter series: CholeskySolver.solve)");
  const size_t B = (size_t)h->B, N = (size_t)h->N, J = (size_t)h->J, R = (size_t)nrhs;
  clr::SweepParams P;
  memset(&P, 0, sizeof(P));
  P.N = h->N; P.J = h->J; P.nrhs = nrhs;
  int nchunk = std::min(clr::wsweep_chunks(h->N, h->J), std::max(2, (int)(2048 / B)));
  if (nchunk > (h->N - 1) / 64) nchunk = std::max(1, (h->N - 1) / 64);
  P.L = (h->N - 1 + nchunk - 1) / nchunk;
  P.nchunk = (h->N - 1 + P.L - 1) / P.L;
  const size_t ws = clr::wsweep_workspace_doubles(h->J, P.nchunk, nrhs);
  if ((st = h->bs_M.reserve(B * ws)) != CLR_OK) return st;
  if ((st = h->bs_x.reserve(B * R * N)) != CLR_OK) return st;   // the forward sweep's output (undivided)
  if ((st = h->bs_rm.reserve(B * R * N)) != CLR_OK) return st;  // right-hand sides in, results out
  const double* src = h->y.p;
  long src_stride = h->y_stride;
  if (b) {
    HIP_TRY(hipMemcpyAsync(h->bs_rm.p, b, B * R * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    src = h->bs_rm.p;
    src_stride = (long)(R * N);
  }
  for (hipEvent_t& e : h->bs_ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipEventRecord(h->bs_ev[0], h->stream));
  P.phi = h->phi.p; P.u = h->u.p; P.W = h->W.p; P.D = h->D.p;
  P.batch = h->B;
  P.stride_phi = (long)(J * (N - 1)); P.stride_W = (long)(J * N); P.stride_D = (long)N;
  P.stride_ws = (long)ws;
  P.in = src; P.stride_in = src_stride;
  P.out = h->bs_x.p; P.stride_out = (long)(R * N);
  P.backward = 0;
  clr::launch_wsweep_scan(P, h->bs_M.p, h->stream);
  P.in = h->bs_x.p; P.stride_in = (long)(R * N);
  P.out = h->bs_rm.p; P.stride_out = (long)(R * N);
  P.backward = 1;
  clr::launch_wsweep_scan(P, h->bs_M.p, h->stream);
  HIP_TRY(hipEventRecord(h->bs_ev[1], h->stream));
  HIP_TRY(hipGetLastError());
  if (x) HIP_TRY(hipMemcpyAsync(x, h->bs_rm.p, B * R * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));  // (null: the result stays in bs_rm)
  HIP_TRY(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->bs_ev[0], h->bs_ev[1]));
  h->solve_device_ms = ms;
  return CLR_OK;
}

// x == null: the result stays on the device, row-major [B][nrhs][N] in bs_rm (clr_batch_predict)
static int batch_solve_impl(clr_batch* h, int nrhs, const double* b, double* x) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (nrhs < 1) return fail(CLR_INVALID_ARGUMENT, "clr_batch_solve: nrhs >= 1");
  if (!b && nrhs != 1) return fail(CLR_INVALID_ARGUMENT, "clr_batch_solve: b == NULL means the plan's own y (one right-hand side)");
  if (!h->launch) return wide_batch_solve(h, nrhs, b, x);
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  if (!h->have_factor || !h->factor_valid) return fail(CLR_NOT_COMPUTED, "no materialising run has been made (clr_batch_enqueue(h, 1))");
  if (h->factor_is_lean && h->factor_inputs_changed)
    return fail(CLR_NOT_COMPUTED, "the lean factor's phi and u are regenerated from the plan's series and coefficients, "
                                  "which were replaced after the materialising run: materialise again");
  if (h->nchunk < 2) return fail(CLR_UNSUPPORTED, "clr_batch_solve needs a chunked plan (N >= 128)");
  clr::BatchParams P;
  if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
  if (P.staged) {  // the row-major times addressed directly (lean: one time per step and lane)
    P.t = h->t.p; P.t_stride = h->t_stride; P.lane_is = 1; P.lane_cs = h->L; P.staged = 0;
  }
  const size_t B = (size_t)h->B, N = (size_t)h->N, J = (size_t)h->J, R = (size_t)nrhs, cells = (size_t)h->L * h->nchunk;
  if ((st = h->bs_x.reserve(B * R * cells)) != CLR_OK) return st;
  if ((st = h->bs_rm.reserve(B * R * N)) != CLR_OK) return st;
  if ((st = h->bs_M.reserve(B * h->nchunk * J * J)) != CLR_OK) return st;
  if ((st = h->bs_off.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
  if ((st = h->bs_starts.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
  const double* src = h->y.p;
  long src_stride = h->y_stride;
  if (b) {
    HIP_TRY(hipMemcpyAsync(h->bs_rm.p, b, B * R * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    src = h->bs_rm.p;
    src_stride = (long)N;
  }
  for (hipEvent_t& e : h->bs_ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  hipEvent_t e0 = h->bs_ev[0], e1 = h->bs_ev[1];
  HIP_TRY(hipEventRecord(e0, h->stream));
  clr::launch_relayout(src, src_stride, h->bs_x.p, (long)cells, (int)(B * R), h->N, h->L, h->nchunk, 0, h->stream);
  clr::BSolveParams S;
  S.nrhs = nrhs; S.r = 0; S.lean = h->factor_is_lean ? 1 : 0;
  // (the chunk maps depend on the factor only: formed by the first solve after a materialising run; they count as
  //  formed only once that solve's kernels have run to completion -- a failed launch or copy must not leave the next
  //  solve reading uninitialised maps)
  S.have_M = h->bs_M_valid ? 1 : 0;
  h->bs_M_valid = false;
  S.xT = h->bs_x.p; S.M = h->bs_M.p; S.off = h->bs_off.p; S.starts = h->bs_starts.p;
  h->launch->bsolve(P, S, h->stream);
  clr::launch_relayout_back(h->bs_x.p, (long)cells, h->bs_rm.p, (long)N, (int)(B * R), h->N, h->L, h->nchunk, h->stream);
  HIP_TRY(hipEventRecord(e1, h->stream));
  HIP_TRY(hipGetLastError());
  if (x) HIP_TRY(hipMemcpyAsync(x, h->bs_rm.p, B * R * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->bs_M_valid = true;
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  h->solve_device_ms = ms;
  return CLR_OK;
}

int clr_batch_solve(clr_batch* h, int nrhs, const double* b, double* x) {
  if (!x) return fail(CLR_INVALID_ARGUMENT, "clr_batch_solve: an output array");
  return batch_solve_impl(h, nrhs, b, x);
}

// CholeskySolver::dot_L (cholesky.h:409-431; GP.sample draws L z, celerite.py:422-451) for every problem of the plan from
// the factor of its last materialising run.  Narrow plans: the chunked diagonal scan of clr_bdotl_kernels.h on the
// chunk-interleaved factor (either layout), lane = (problem, chunk).  Wide plans (widths 9..64): the factor lies in the
// reference's storage and the object API's wave-per-chunk scan (wsweep_kernels.hip: wdotl_kernel) runs once for the whole
// batch, grid.z = problem; series too short for it: the sequential kernel, problem by problem.
int clr_batch_dot_L(clr_batch* h, int nrhs, const double* z, double* y) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (nrhs < 1 || nrhs > 65535 || !z || !y) return fail(CLR_INVALID_ARGUMENT, "clr_batch_dot_L: 1 <= nrhs <= 65535 (grid.z), z and an output array");
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  if (!h->have_factor || !h->factor_valid) return fail(CLR_NOT_COMPUTED, "no materialising run has been made (clr_batch_enqueue(h, 1))");
  if (h->J_general > 0 || h->J > clr::wide_max_width()) return fail(CLR_UNSUPPORTED, "clr_batch_dot_L covers celerite-only plans of widths 1..64");
  const size_t B = (size_t)h->B, N = (size_t)h->N, J = (size_t)h->J, R = (size_t)nrhs;
  if ((st = h->bs_rm.reserve(B * R * N)) != CLR_OK) return st;
  for (hipEvent_t& e : h->bs_ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipMemcpyAsync(h->bs_rm.p, z, B * R * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const double* result = nullptr;
  if (!h->launch) {  // wide plans
    if ((st = h->bs_x.reserve(B * R * N)) != CLR_OK) return st;
    HIP_TRY(hipEventRecord(h->bs_ev[0], h->stream));
    if (clr::wdotl_scan_supported(h->N, h->J)) {
      clr::SweepParams P;
      memset(&P, 0, sizeof(P));
      P.N = h->N; P.J = h->J; P.nrhs = nrhs;
      // chunks per problem: about two rounds of the chip's SIMDs over the batch (not a function of nrhs: a right-hand
      // side's result does not depend on how many others ride along)
      int nchunk = std::min(clr::wdotl_chunks(h->N), std::max(2, (int)(2048 / B)));
      P.L = (h->N - 1 + nchunk - 1) / nchunk;
      P.nchunk = (h->N - 1 + P.L - 1) / P.L;
      const size_t ws = R * (size_t)P.nchunk * 3 * J;
      if ((st = h->bs_off.reserve(B * ws)) != CLR_OK) return st;
      P.phi = h->phi.p; P.u = h->u.p; P.W = h->W.p; P.D = h->D.p;
      P.batch = h->B;
      P.stride_phi = (long)(J * (N - 1)); P.stride_W = (long)(J * N); P.stride_D = (long)N;
      P.stride_ws = (long)ws;
      P.in = h->bs_rm.p; P.stride_in = (long)(R * N);
      P.out = h->bs_x.p; P.stride_out = (long)(R * N);
      clr::launch_wdotl_scan(P, h->bs_off.p, h->stream);
    } else {
      for (size_t p = 0; p < B; ++p)
        clr::launch_dot_L(h->N, h->J, nrhs, h->phi.p + p * J * (N - 1), h->u.p + p * J * (N - 1), h->W.p + p * J * N,
                          h->D.p + p * N, h->bs_rm.p + p * R * N, h->bs_x.p + p * R * N, h->stream);
    }
    HIP_TRY(hipEventRecord(h->bs_ev[1], h->stream));
    result = h->bs_x.p;
  } else {
    if (h->factor_is_lean && h->factor_inputs_changed)
      return fail(CLR_NOT_COMPUTED, "the lean factor's phi and u are regenerated from the plan's series and coefficients, "
                                    "which were replaced after the materialising run: materialise again");
    clr::BatchParams P;
    if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
    if (P.staged) {  // the row-major times addressed directly (lean: one time per step and lane)
      P.t = h->t.p; P.t_stride = h->t_stride; P.lane_is = 1; P.lane_cs = h->L; P.staged = 0;
    }
    const size_t cells = (size_t)h->L * h->nchunk;
    if ((st = h->bs_x.reserve(B * R * cells)) != CLR_OK) return st;
    if ((st = h->bs_decay.reserve(B * h->nchunk * J)) != CLR_OK) return st;
    if ((st = h->bs_off.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
    if ((st = h->bs_starts.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
    HIP_TRY(hipEventRecord(h->bs_ev[0], h->stream));
    clr::launch_relayout(h->bs_rm.p, (long)N, h->bs_x.p, (long)cells, (int)(B * R), h->N, h->L, h->nchunk, 0, h->stream);
    clr::BDotLParams S;
    S.nrhs = nrhs; S.lean = h->factor_is_lean ? 1 : 0;
    S.xT = h->bs_x.p; S.decay = h->bs_decay.p; S.off = h->bs_off.p; S.starts = h->bs_starts.p;
    h->launch->bdotl(P, S, h->stream);
    clr::launch_relayout_back(h->bs_x.p, (long)cells, h->bs_rm.p, (long)N, (int)(B * R), h->N, h->L, h->nchunk, h->stream);
    HIP_TRY(hipEventRecord(h->bs_ev[1], h->stream));
    result = h->bs_rm.p;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(y, result, B * R * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->bs_ev[0], h->bs_ev[1]));
  h->solve_device_ms = ms;
  return CLR_OK;
}

// the constant diagonal of one problem's K for dot: sum a_real + sum a_comp + jitter (cholesky.h:483-485), N copies
__global__ void __launch_bounds__(256) dot_diagonal_kernel(const double* a_real, const double* a_comp, const double* jitter, int JR,
                                                           int JC, double* dg, int N) {
  double sr = 0.0, sc = 0.0;
  for (int j = 0; j < JR; ++j) sr += a_real[j];
  for (int j = 0; j < JC; ++j) sc += a_comp[j];
  const double d = (sr + sc) + jitter[0];
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) dg[n] = d;
}

// CholeskySolver::dot (cholesky.h:441-596; GP.dot, celerite.py:453-489) for every problem of the plan: y_p = K_p z_p with
// K_p given by the plan's resident times and the coefficients in force (diagonal sum a_real + sum a_comp + jitter: no
// observational variance, :483-485) -- no factor, no materialising run.  Narrow plans: the two triangles as chunked
// diagonal scans with the features evaluated on the fly (clr_bdot_kernels.h), lane = (problem, chunk).  Wide plans
// (widths 9..64): the object API's kernels problem by problem (launch_dot_setup + the wave-per-chunk scans of
// wsweep_kernels.hip, or the sequential kernel on short series) on the plan's resident arrays.
int clr_batch_dot(clr_batch* h, int nrhs, const double* z, double* y) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (nrhs < 1 || nrhs > 65535 || !z || !y) return fail(CLR_INVALID_ARGUMENT, "clr_batch_dot: 1 <= nrhs <= 65535 (grid.z), z and an output array");
  if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st;
  if (h->J_general > 0 || h->J > clr::wide_max_width()) return fail(CLR_UNSUPPORTED, "clr_batch_dot covers celerite-only plans of widths 1..64");
  clr::BatchParams P;
  if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
  const size_t B = (size_t)h->B, N = (size_t)h->N, J = (size_t)h->J, R = (size_t)nrhs;
  if ((st = h->bs_rm.reserve(B * R * N)) != CLR_OK) return st;
  for (hipEvent_t& e : h->bs_ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipMemcpyAsync(h->bs_rm.p, z, B * R * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const double* result = nullptr;
  if (!h->launch) {  // wide plans: problem by problem
    DevBuf feat, dgb, ws;
    auto cleanup = [&](int code) { (void)hipStreamSynchronize(h->stream); feat.release(); dgb.release(); ws.release(); return code; };
    if ((st = feat.reserve(3 * J * N)) != CLR_OK) return cleanup(st);
    if ((st = dgb.reserve(N)) != CLR_OK) return cleanup(st);  // the problem's constant diagonal, refilled per problem (same stream)
    if ((st = h->bs_x.reserve(B * R * N)) != CLR_OK) return cleanup(st);
    const bool scan = clr::wdotl_scan_supported(h->N, h->J);
    clr::SweepParams SP;
    memset(&SP, 0, sizeof(SP));
    if (scan) {
      SP.N = h->N; SP.J = h->J; SP.nrhs = nrhs;
      SP.nchunk = clr::wdotl_chunks(h->N);
      SP.L = (h->N - 1 + SP.nchunk - 1) / SP.nchunk;
      SP.nchunk = (h->N - 1 + SP.L - 1) / SP.L;
      if ((st = ws.reserve(R * (size_t)SP.nchunk * 3 * J)) != CLR_OK) return cleanup(st);
    }
    (void)hipEventRecord(h->bs_ev[0], h->stream);
    for (size_t p = 0; p < B; ++p) {
      clr::GenericProblem g;
      g.N = h->N; g.J = h->J; g.J_real = h->J_real; g.J_comp = h->J_comp; g.J_general = 0;
      g.a_real = P.a_real + p * h->J_real; g.c_real = P.c_real + p * h->J_real;
      g.a_comp = P.a_comp + p * h->J_comp; g.b_comp = P.b_comp + p * h->J_comp;
      g.c_comp = P.c_comp + p * h->J_comp; g.d_comp = P.d_comp + p * h->J_comp;
      g.U = nullptr; g.V = nullptr;
      g.t = h->t.p + p * (size_t)h->t_stride;
      double *phi = feat.p, *u = feat.p + J * N, *v = feat.p + 2 * J * N;
      clr::launch_dot_setup(g, phi, u, v, h->stream);
      hipLaunchKernelGGL(dot_diagonal_kernel, dim3((unsigned)std::min<size_t>((N + 255) / 256, 1024)), dim3(256), 0, h->stream,
                         g.a_real, g.a_comp, P.jitter + p, h->J_real, h->J_comp, dgb.p, h->N);
      if (scan) {
        SP.phi = phi; SP.u = u;
        SP.in = h->bs_rm.p + p * R * N; SP.out = h->bs_x.p + p * R * N;
        clr::launch_wdot_scan(SP, v, dgb.p, ws.p, h->stream);
      } else {
        clr::launch_dot(h->N, h->J, nrhs, phi, u, v, dgb.p, h->bs_rm.p + p * R * N, h->bs_x.p + p * R * N, h->stream);
      }
    }
    (void)hipEventRecord(h->bs_ev[1], h->stream);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(y, h->bs_x.p, B * R * N * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess)
      return cleanup(fail(CLR_HIP_ERROR, "clr_batch_dot: kernels or the download failed"));
    if ((st = cleanup(CLR_OK)) != CLR_OK) return st;
  } else {
    // the row-major times addressed directly (the chunk-interleaved copy of the role-split summarize is only made by an
    // evaluation: clr_batch_dot must not depend on one having run)
    P.t = h->t.p; P.t_stride = h->t_stride; P.lane_is = 1; P.lane_cs = h->L; P.staged = 0;
    const size_t cells = (size_t)h->L * h->nchunk;
    if ((st = h->bs_x.reserve(B * R * cells)) != CLR_OK) return st;
    if ((st = h->bs_y.reserve(B * R * cells)) != CLR_OK) return st;
    if ((st = h->bs_decay.reserve(B * h->nchunk * J)) != CLR_OK) return st;
    if ((st = h->bs_off.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
    if ((st = h->bs_starts.reserve(B * R * h->nchunk * J)) != CLR_OK) return st;
    HIP_TRY(hipEventRecord(h->bs_ev[0], h->stream));
    clr::launch_relayout(h->bs_rm.p, (long)N, h->bs_x.p, (long)cells, (int)(B * R), h->N, h->L, h->nchunk, 0, h->stream);
    clr::BDotParams S;
    S.nrhs = nrhs; S.zT = h->bs_x.p; S.yT = h->bs_y.p; S.decay = h->bs_decay.p; S.off = h->bs_off.p; S.starts = h->bs_starts.p;
    h->launch->bdot(P, S, h->stream);
    clr::launch_relayout_back(h->bs_y.p, (long)cells, h->bs_rm.p, (long)N, (int)(B * R), h->N, h->L, h->nchunk, h->stream);
    HIP_TRY(hipEventRecord(h->bs_ev[1], h->stream));
    result = h->bs_rm.p;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(y, result, B * R * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->bs_ev[0], h->bs_ev[1]));
  h->solve_device_ms = ms;
  return CLR_OK;
}

// CholeskySolver::predict (cholesky.h:599-698; GP.predict's conditional mean, celerite.py:330-420) for every problem of
// the plan: mu*_p(x*) = K_p(x*, t_p) K_p^-1 y_p at M points per problem.  alpha = K^-1 y is the batched solve on the
// materialised factor (either layout, any width <= 64), left on the device; the reference's two passes over the sorted
// prediction points are the object API's chunked diagonal scans (generic_kernels.hip: launch_predict_scan, a parallel
// prefix over chunks of 16 samples + one thread per point) run problem by problem on the plan's resident times and
// coefficients -- they need nothing of the factor.  Unsorted points: the sequential walk (launch_predict).
int clr_batch_predict(clr_batch* h, int M, const double* xs, long xs_stride, double* pred) {
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (M < 0 || (M > 0 && (!xs || !pred))) return fail(CLR_INVALID_ARGUMENT, "clr_batch_predict: M >= 0, the points and an output array");
  if (xs_stride != 0 && xs_stride != M) return fail(CLR_INVALID_ARGUMENT, "clr_batch_predict: the points' stride is 0 (shared by all problems) or M");
  if (h->J_general > 0 || h->J > clr::wide_max_width()) return fail(CLR_UNSUPPORTED, "clr_batch_predict covers celerite-only plans of widths 1..64");
  if (M == 0) return CLR_OK;
  if ((st = batch_solve_impl(h, 1, nullptr, nullptr)) != CLR_OK) return st;  // alpha = K^-1 y -> bs_rm [B][N]
  clr::BatchParams P;
  if ((st = batch_params(h, 0, P)) != CLR_OK) return st;
  const size_t B = (size_t)h->B, Mm = (size_t)M, nsrc = xs_stride == 0 ? 1 : B;
  DevBuf dxs, dpred, ws;
  auto cleanup = [&](int code) { dxs.release(); dpred.release(); ws.release(); return code; };
  if ((st = dxs.reserve(nsrc * Mm)) != CLR_OK) return cleanup(st);
  if ((st = dpred.reserve(B * Mm)) != CLR_OK) return cleanup(st);
  const bool scan_ok = clr::predict_scan_supported(h->N, h->J_real, h->J_comp);
  int pchunk = 0, pL = 0;
  if (scan_ok) {
    pchunk = std::max(1, std::min(h->N / 16, 8192));
    pL = (h->N + pchunk - 1) / pchunk;
    pchunk = (h->N + pL - 1) / pL;
    if ((st = ws.reserve(clr::predict_workspace_doubles(pchunk, h->J))) != CLR_OK) return cleanup(st);
  }
  if (hipMemcpyAsync(dxs.p, xs, nsrc * Mm * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemsetAsync(dpred.p, 0, B * Mm * sizeof(double), h->stream) != hipSuccess)
    return cleanup(fail(CLR_HIP_ERROR, "clr_batch_predict: upload failed"));
  std::vector<char> sorted(nsrc, 1);
  for (size_t p = 0; p < nsrc; ++p)
    for (size_t m = 1; m < Mm && sorted[p]; ++m) sorted[p] = xs[p * Mm + m - 1] <= xs[p * Mm + m];
  for (size_t p = 0; p < B; ++p) {
    clr::GenericProblem g;
    g.N = h->N; g.J = h->J; g.J_real = h->J_real; g.J_comp = h->J_comp; g.J_general = 0;
    g.a_real = P.a_real + p * h->J_real; g.c_real = P.c_real + p * h->J_real;
    g.a_comp = P.a_comp + p * h->J_comp; g.b_comp = P.b_comp + p * h->J_comp;
    g.c_comp = P.c_comp + p * h->J_comp; g.d_comp = P.d_comp + p * h->J_comp;
    g.U = nullptr; g.V = nullptr;
    g.t = h->t.p + p * (size_t)h->t_stride;
    const double* alpha = h->bs_rm.p + p * (size_t)h->N;
    const double* xp = dxs.p + (xs_stride == 0 ? 0 : p * Mm);
    if (scan_ok && sorted[xs_stride == 0 ? 0 : p]) clr::launch_predict_scan(g, alpha, M, xp, dpred.p + p * Mm, ws.p, pchunk, pL, h->stream);
    else clr::launch_predict(g, alpha, M, xp, dpred.p + p * Mm, h->stream);
  }
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(pred, dpred.p, B * Mm * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return cleanup(fail(CLR_HIP_ERROR, "clr_batch_predict: kernels or the download failed"));
  return cleanup(CLR_OK);
}

int clr_batch_get_solve_ms(const clr_batch* h, double* device_ms) {
  if (device_ms) *device_ms = h->solve_device_ms;
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
  h->pin_results = false;
  if (h->J_general > 0 || h->J > clr::wide_max_width()) {
    // (plans on the any-width sequential kernel or with general terms: the evaluation itself, `steps` times, as one "replay" slot)
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, h->stream));
    for (int i = 0; i < steps; ++i)
      if ((st = clr_batch_enqueue(h, materialize)) != CLR_OK) return st;
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float tot = 0.f;
    HIP_TRY(hipEventElapsedTime(&tot, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (total_ms) *total_ms = tot;
    if (kernel_ms) { for (int j = 0; j < 6; ++j) kernel_ms[j] = 0.0; kernel_ms[4] = tot; }
    return CLR_OK;
  }
  if (!warm_runs(h, materialize) && h->relayout_pending && !relayout_each_step && batch_relayout(h)) h->relayout_pending = false;
  // one event per kernel boundary per step, all recorded on the handle's stream
  const int NK = 6;
  std::vector<hipEvent_t> ev((size_t)steps * (NK + 1));
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < steps; ++i) {
    hipEvent_t* e = &ev[(size_t)i * (NK + 1)];
    HIP_TRY(hipEventRecord(e[0], h->stream));
    if (!h->launch) {  // wide path (one chunk: the whole sweep is reported in the "replay" slot)
      if ((st = wide_launch(h, P, e)) != CLR_OK) return st;
      if (materialize) { h->factor_is_lean = false; h->factor_inputs_changed = false; h->factor_valid = true; h->bs_M_valid = false; }
      // (a plan that re-planned level-1 problems at its last evaluation does so inside every timed step: the step's
      //  time then includes the side plan -- at the price of a host round trip per step)
      if (P.defer_level1 && h->rescue_last != 0) { h->rescue_inflight = true; if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st; HIP_TRY(hipEventRecord(e[6], h->stream)); }
      else h->rescue_inflight = P.defer_level1 != 0;
      continue;
    }
    if (!warm_runs(h, materialize) && small_runs(h, materialize)) {  // (one launch, in the "summarize" slot)
      clr::BatchParams Sp;
      if (relayout_each_step) h->small_copy_pending = true;  // (new series every step: the copy is rebuilt inside it)
      if ((st = small_params(h, Sp)) != CLR_OK) return st;
      HIP_TRY(hipEventRecord(e[1], h->stream));
      clr::launch_small_batch(h->J_real, h->J_comp, Sp, 256, h->stream);
      for (int j = 2; j <= 6; ++j) HIP_TRY(hipEventRecord(e[j], h->stream));
      h->small_inflight = true;
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
    if (mp_runs(h, materialize)) {  // (the whole pipeline in the "replay" slot)
      for (int j = 1; j <= 4; ++j) HIP_TRY(hipEventRecord(e[j], h->stream));
      if ((st = materialize_pipeline(h, P)) != CLR_OK) return st;
      HIP_TRY(hipEventRecord(e[5], h->stream));
      HIP_TRY(hipEventRecord(e[6], h->stream));
      continue;
    }
    HIP_TRY(hipEventRecord(e[1], h->stream));
    h->launch->summarize(P, h->stream);
    HIP_TRY(hipEventRecord(e[2], h->stream));
    h->launch->prefix(P, h->stream);
    HIP_TRY(hipEventRecord(e[3], h->stream));
    h->launch->correct(P, h->stream);
    HIP_TRY(hipEventRecord(e[4], h->stream));
    h->launch->replay(replay_view(h, P, materialize), replay_mode(h, materialize), h->stream);
    if (P.ends) refine_chunk_heads(h, replay_view(h, P, materialize), materialize, h->stream);
    h->launch->sequential(P, replay_mode(h, materialize), h->stream);
    if (materialize) { h->factor_is_lean = h->factor_layout == 1; h->factor_inputs_changed = false; h->factor_valid = true; h->bs_M_valid = false; }
    HIP_TRY(hipEventRecord(e[5], h->stream));
    clr::launch_finalize(P, h->stream);
    if (P.defer_level1 && h->rescue_last != 0) { h->rescue_inflight = true; if ((st = warm_resolve(h, nullptr)) != CLR_OK) return st; }
    else h->rescue_inflight = P.defer_level1 != 0;
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

// one optimiser / MCMC evaluation in one call: new coefficients in, the B results out (celerite.py:160-219 per problem:
// set_parameter_vector -> compute -> log_likelihood)
int clr_batch_evaluate(clr_batch* h, const double* jitter, const double* a_real, const double* c_real,
                       const double* a_comp, const double* b_comp, const double* c_comp, const double* d_comp,
                       double* loglike, double* logdet, double* quad, int* status) {
  int st = clr_batch_set_coefficients(h, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp);
  if (st == CLR_OK) st = clr_batch_enqueue(h, 0);
  if (st == CLR_OK) st = clr_batch_get_results(h, loglike, logdet, quad, status);
  return st;
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
    if (width < 1 || width > CLR_MAX_WIDTH) return CLR_UNSUPPORTED;
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

}  // extern "C"

// ---- hooks of the sharded layer (clr_group_hooks.h; not part of the C ABI) ----------------------------------------
namespace clr_group {

void set_batch_context(clr_batch* h, int B_total) {
  h->group_B = B_total > h->B ? B_total : 0;  // (a single shard IS the whole batch)
}

long warm_eligible(const clr_batch* h) { return h->warm_eligible; }

void set_warm_eligible_total(clr_batch* h, long eligible_total) {
  if (h->group_B <= 0) return;
  const bool selectable = h->warm_mode != 0 && h->wnchunk >= 2 && h->have_series && !h->warm_span.empty() &&
                          h->host_cmin.size() == (size_t)h->B && h->warm_K.size() == (size_t)h->B;
  h->warm_eligible_total = eligible_total;
  h->warm_active = selectable && eligible_total * 2 >= (long)h->group_B;
  h->warm_K_dirty = h->warm_active;  // (K per problem goes up with the next enqueue)
}

bool in_flight(const clr_batch* h) { return h->res_open || h->warm_inflight || h->small_inflight || h->rescue_inflight; }

int resolve_begin(clr_batch* h, long* pending, long* eligible) {
  const int st = require_device(h->device);
  if (st != CLR_OK) return st;
  return ::resolve_begin(h, pending, eligible);
}

int resolve_finish(clr_batch* h, long pending_total, long eligible_total) {
  const int st = require_device(h->device);
  if (st != CLR_OK) return st;
  return ::resolve_finish(h, pending_total, eligible_total);
}

}  // namespace clr_group
