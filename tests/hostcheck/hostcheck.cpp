// tests/hostcheck/hostcheck.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host (g++) instantiation of the per-lane device templates in
// celerite_amd/csrc/clr_core.h, with the GPU grid replaced by plain loops, so
// the scan algebra (summarize -> prefix -> replay) can be checked against the
// oracle in the CPU-only test run.  Not linked into libcelerite_hip.so, never
// used by the product: the shipped library has no CPU implementation.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../celerite_amd/csrc/clr_core.h"

using namespace clr;

// prefix schedule of the next hostcheck_batch calls: 0 levels = the plain walk over the chunks;
// otherwise the multi-level prefix (compose groups bottom-up, walk the top, fan out), group size g
static int g_prefix_levels = 0, g_prefix_g = 0;
// diagnostics of the last hostcheck_batch call, per problem: gamma_max, mu_min, realized G error (max over chunks),
// and never_replay: report the chunk-summary (route 0) values even for flagged problems
static std::vector<double> g_diag;
static int g_never_replay = 0;
extern "C" void hostcheck_set_never_replay(int on) { g_never_replay = on; }
extern "C" int hostcheck_get_diag(int B, double* out /* [B][3] */) {
  if ((int)g_diag.size() < 3 * B) return -1;
  memcpy(out, g_diag.data(), sizeof(double) * 3 * B);
  return 0;
}
extern "C" void hostcheck_set_prefix(int levels, int g) { g_prefix_levels = levels; g_prefix_g = g; }

// start states of all `n` elements of one level from the level's elements (ELEM doubles each):
// recursion over groups of g
template <int J>
static void multilevel_starts(const std::vector<double>& elems, int n, int levels, int g,
                              std::vector<double>& starts) {
  constexpr int SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  starts.assign((size_t)n * START, 0.0);
  double dld, dq;
  int sus;
  if (levels == 0 || n < 2) {
    double S[SZ] = {0}, f[J] = {0};
    for (int c = 0; c + 1 < n; ++c) {
      chunk_update<J>(&elems[(size_t)c * ELEM], S, f, false, true, 0.0, 0.0, &dld, &dq, &sus);
      memcpy(&starts[(size_t)(c + 1) * START], S, sizeof(S));
      memcpy(&starts[(size_t)(c + 1) * START + SZ], f, sizeof(f));
    }
    return;
  }
  const int np = (n + g - 1) / g;
  std::vector<double> parents((size_t)np * ELEM), pstarts;
  for (int k = 0; k < np; ++k) {
    const int first = k * g, len = std::min(g, n - first);
    double e[ELEM];
    memcpy(e, &elems[(size_t)first * ELEM], sizeof(e));
    for (int i = 1; i < len; ++i) compose_elements<J>(e, &elems[(size_t)(first + i) * ELEM], e);
    memcpy(&parents[(size_t)k * ELEM], e, sizeof(e));
  }
  multilevel_starts<J>(parents, np, levels - 1, g, pstarts);
  for (int k = 0; k < np; ++k) {
    const int first = k * g, len = std::min(g, n - first);
    double S[SZ], f[J];
    memcpy(S, &pstarts[(size_t)k * START], sizeof(S));
    memcpy(f, &pstarts[(size_t)k * START + SZ], sizeof(f));
    for (int i = 0; i < len; ++i) {
      memcpy(&starts[(size_t)(first + i) * START], S, sizeof(S));
      memcpy(&starts[(size_t)(first + i) * START + SZ], f, sizeof(f));
      if (i + 1 < len)
        chunk_update<J>(&elems[(size_t)(first + i) * ELEM], S, f, false, true, 0.0, 0.0, &dld, &dq, &sus);
    }
  }
}

template <int JR, int JC, bool FAST>
static int run(int B, int N, int nchunk, const double* jitter, const double* a_real,
               const double* c_real, const double* a_comp, const double* b_comp,
               const double* c_comp, const double* d_comp, const double* t, long ts,
               const double* diag, long ds, const double* y, long ys, int materialize,
               int interleaved, int exact,
               double* ll, double* logdet, double* quad, int* status, double* phi, double* u,
               double* W, double* D, int* used_exact) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  const int L = (N + nchunk - 1) / nchunk;
  std::vector<double> elems((size_t)nchunk * Wd::ELEM), starts((size_t)nchunk * Wd::START);
  // optional chunk-interleaved copies [i][chunk] (what relayout_kernel writes on the GPU)
  std::vector<double> tT((size_t)L * nchunk), dT((size_t)L * nchunk), yT((size_t)L * nchunk);
  g_diag.assign((size_t)3 * B, 0.0);
  for (int b = 0; b < B; ++b) {
    Problem<JR, JC> p;
    p.load(a_real + (long)b * JR, c_real + (long)b * JR, a_comp + (long)b * JC,
           b_comp + (long)b * JC, c_comp + (long)b * JC, d_comp + (long)b * JC, jitter[b]);
    const double *tb = t + b * ts, *db = diag + b * ds, *yb = y + b * ys;
    long is = 1, cs = L;
    if (interleaved) {
      for (int c = 0; c < nchunk; ++c)
        for (int i = 0; i < L; ++i) {
          const long n = (long)c * L + i;
          tT[(size_t)i * nchunk + c] = n < N ? tb[n] : 0.0;
          dT[(size_t)i * nchunk + c] = n < N ? db[n] : 0.0;
          yT[(size_t)i * nchunk + c] = n < N ? yb[n] : 0.0;
        }
      tb = tT.data(); db = dT.data(); yb = yT.data();
      is = nchunk; cs = 1;
    }
    auto lane = [&](int c) {
      return DirectSeries{tb + c * cs, db + c * cs, yb + c * cs, is, cs, L, (long)N - (long)c * L};
    };
    // summarize every chunk that has data (zero-start sums + element)
    std::vector<double> ld0(nchunk, 0.0), q0(nchunk, 0.0);
    std::vector<int> fl0(nchunk, 0);
    int nreal = 0;
    for (int c = 0; c < nchunk; ++c) {
      if ((long)c * L >= N) break;
      nreal = c + 1;
      DirectSeries src = lane(c);
      double gam = 0.0;
      summarize_chunk<JR, JC, FAST>(p, src, L, c * L, N, true, &elems[(size_t)c * Wd::ELEM], &ld0[c],
                                    &q0[c], &fl0[c], &gam);
      if (!(gam <= g_diag[3 * b])) g_diag[3 * b] = gam;
    }
    g_diag[3 * b + 1] = 1.0;
    // prefix: corrections with the incoming state, then advance
    double S[Wd::SZ] = {0}, f[J] = {0};
    double ld = 0, qd = 0;
    int need_exact = 0;
    std::vector<double> mstarts;
    if (g_prefix_levels > 0) {
      std::vector<double> real_elems(elems.begin(), elems.begin() + (size_t)nreal * Wd::ELEM);
      multilevel_starts<J>(real_elems, nreal, g_prefix_levels, g_prefix_g, mstarts);
    }
    for (int c = 0; c < nreal; ++c) {
      double dld = 0, dq = 0;
      int sus = 0;
      if (g_prefix_levels > 0) {  // the state this chunk starts from, as the multi-level prefix found it
        memcpy(S, &mstarts[(size_t)c * Wd::START], sizeof(S));
        memcpy(f, &mstarts[(size_t)c * Wd::START + Wd::SZ], sizeof(f));
      }
      // chunk 0 starts from the zero state: no correction (E = I)
      double mu = 1.0, eg = 0.0;
      chunk_update<J>(&elems[(size_t)c * Wd::ELEM], S, f, c > 0, c + 1 < nreal, ld0[c], q0[c], &dld,
                      &dq, &sus, &mu, true, c > 0 ? &eg : nullptr);
      if (c > 0 && !(mu >= g_diag[3 * b + 1])) g_diag[3 * b + 1] = mu;
      if (!(eg <= g_diag[3 * b + 2])) g_diag[3 * b + 2] = eg;
      ld += ld0[c] + dld;
      qd += q0[c] + dq;
      need_exact |= sus | fl0[c];
      if (c + 1 < nreal) {
        if (g_prefix_levels > 0) {
          memcpy(&starts[(size_t)(c + 1) * Wd::START], &mstarts[(size_t)(c + 1) * Wd::START],
                 sizeof(double) * Wd::START);
        } else {
          memcpy(&starts[(size_t)(c + 1) * Wd::START], S, sizeof(S));
          memcpy(&starts[(size_t)(c + 1) * Wd::START + Wd::SZ], f, sizeof(f));
        }
      }
    }
    int bad = 0;
    const int flagged = need_exact;
    if (g_never_replay) need_exact = 0;
    if (exact || need_exact || materialize) {  // the exact replay (reference semantics)
      ld = 0;
      qd = 0;
      for (int c = 0; c < nreal; ++c) {
        const int n0 = c * L;
        double l, q;
        int fl;
        const long Nm1 = N - 1;
        DirectSeries src = lane(c);
        if (materialize)
          replay_chunk<JR, JC, 1, FAST>(p, src, L, N, n0, c ? &starts[(size_t)c * Wd::START] : nullptr,
                                        &l, &q, &fl, phi + (long)b * J * Nm1, u + (long)b * J * Nm1,
                                        W + (long)b * J * N, D + (long)b * N, 0);
        else
          replay_chunk<JR, JC, 0, FAST>(p, src, L, N, n0, c ? &starts[(size_t)c * Wd::START] : nullptr,
                                        &l, &q, &fl, nullptr, nullptr, nullptr, nullptr, 0);
        ld += l;
        qd += q;
        bad |= fl;
      }
    }
    used_exact[b] = g_never_replay ? flagged : ((exact || need_exact || materialize) ? 1 : 0);
    status[b] = bad ? 2 : 0;
    logdet[b] = bad ? NAN : ld;
    quad[b] = bad ? NAN : qd;
    ll[b] = bad ? -INFINITY : combine_loglike(ld, qd, N);
  }
  return 0;
}

#define CASE(R, C)                                                                          \
  if (JR == R && JC == C && fast)                                                           \
    return run<R, C, true>(B, N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp,    \
                           d_comp, t, ts, diag, ds, y, ys, materialize, interleaved, exact, \
                           ll, logdet, quad, status, phi, u, W, D, used_exact);             \
  if (JR == R && JC == C)                                                                   \
    return run<R, C, false>(B, N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp,  \
                     t, ts, diag, ds, y, ys, materialize, interleaved, exact, ll, logdet,   \
                     quad, status, phi, u, W, D, used_exact);

extern "C" int hostcheck_batch(int B, int N, int JR, int JC, int nchunk, const double* jitter,
                               const double* a_real, const double* c_real,
                               const double* a_comp, const double* b_comp,
                               const double* c_comp, const double* d_comp, const double* t,
                               long ts, const double* diag, long ds, const double* y, long ys,
                               int materialize, int interleaved, int fast, int exact,
                               double* ll, double* logdet, double* quad,
                               int* status, double* phi, double* u, double* W, double* D,
                               int* used_exact) {
  CASE(1, 0) CASE(2, 0) CASE(3, 0) CASE(0, 1) CASE(1, 1) CASE(2, 1) CASE(0, 2) CASE(2, 2)
  CASE(2, 3) CASE(0, 4) CASE(4, 2) CASE(8, 0)
#ifdef HOSTCHECK_WIDE
  CASE(0, 16)
#endif
  return -1;
}

// The warm-started plain recurrence (clr_batch_kernels.h: warm_kernel + warm_check_kernel) on the host: every chunk
// runs replay_chunk from the zero state K samples before its first sample (chunk 0: from its first sample), the
// state after the warm-up is compared with the state the previous chunk ends in.  Returns per problem log det,
// quadratic form, the largest relative boundary mismatch and whether any pivot was flagged.
template <int JR, int JC>
static int run_warm(int B, int N, int nchunk, int K, const double* jitter, const double* a_real, const double* c_real,
                    const double* a_comp, const double* b_comp, const double* c_comp, const double* d_comp,
                    const double* t, const double* diag, const double* y, double* logdet, double* quad, double* resid,
                    int* flagged) {
  using Wd = Widths<JR, JC>;
  const int L = (N + nchunk - 1) / nchunk;
  for (int b = 0; b < B; ++b) {
    Problem<JR, JC> p;
    p.load(a_real + (long)b * JR, c_real + (long)b * JR, a_comp + (long)b * JC, b_comp + (long)b * JC,
           c_comp + (long)b * JC, d_comp + (long)b * JC, jitter[b]);
    const double *tb = t + (long)b * N, *db = diag + (long)b * N, *yb = y + (long)b * N;
    double ld = 0, qd = 0, res = 0;
    int bad = 0;
    double prev_end[Wd::START];
    for (int c = 0; c * L < N; ++c) {
      const int warm = c == 0 ? 0 : K;
      const long first = (long)c * L - warm;
      if (first < 0) return -2;
      DirectSeries src{tb + first, db + first, yb + first, 1, L + warm, L + warm, (long)N - first};
      double l, q, st[Wd::START], en[Wd::START];
      int fl;
      replay_chunk<JR, JC, 0, true>(p, src, L + warm, N, (int)first, nullptr, &l, &q, &fl, nullptr, nullptr, nullptr,
                                    nullptr, 0, en, warm, st);
      ld += l; qd += q; bad |= fl;
      if (c > 0) {
        double pm = 0, dp = 0, fm = 0, df = 0;
        for (int i = 0; i < Wd::SZ; ++i) { pm = fmax(pm, fabs(prev_end[i])); dp = fmax(dp, fabs(prev_end[i] - st[i])); }
        for (int i = Wd::SZ; i < Wd::START; ++i) { fm = fmax(fm, fabs(prev_end[i])); df = fmax(df, fabs(prev_end[i] - st[i])); }
        double r = pm > 0 ? dp / pm : (dp == 0 ? 0 : INFINITY);
        if (fm > 0) r = fmax(r, df / fm);
        if (!(r <= res)) res = r;
      }
      memcpy(prev_end, en, sizeof(en));
    }
    logdet[b] = ld; quad[b] = qd; resid[b] = res; flagged[b] = bad;
  }
  return 0;
}

extern "C" int hostcheck_warm(int B, int N, int JR, int JC, int nchunk, int K, const double* jitter, const double* a_real,
                              const double* c_real, const double* a_comp, const double* b_comp, const double* c_comp,
                              const double* d_comp, const double* t, const double* diag, const double* y, double* logdet,
                              double* quad, double* resid, int* flagged) {
#define WCASE(R, C) if (JR == R && JC == C) return run_warm<R, C>(B, N, nchunk, K, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, logdet, quad, resid, flagged);
  WCASE(2, 3) WCASE(0, 2) WCASE(1, 1) WCASE(4, 0)
#undef WCASE
  return -1;
}

// The prefix planner (clr_core.h: plan_prefix), as the library calls it.
extern "C" void hostcheck_plan_prefix(int nchunk, int levels, int g, int B, int J, int* out /* levels, g[3], n[4] */,
                                      double* time_us) {
  const PrefixPlan p = plan_prefix(nchunk, levels, g, B, J);
  out[0] = p.levels;
  for (int l = 0; l < 3; ++l) out[1 + l] = p.g[l];
  for (int l = 0; l < 4; ++l) out[4 + l] = p.n[l];
  *time_us = p.time_us;
}

// Accuracy probes for the two device math helpers (host instantiation).
extern "C" void hostcheck_sincos(int n, const double* x, double* s, double* c) {
  for (int i = 0; i < n; ++i) sincos_fast(x[i], s + i, c + i);
}
extern "C" double hostcheck_logprod(int n, const double* d) {
  LogProduct lp;
  lp.init();
  for (int i = 0; i < n; ++i) lp.mul(d[i]);
  return lp.log_value();
}
