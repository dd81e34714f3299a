// tests/hostcheck/hostcheck_grad.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host (g++) instantiation of celerite_amd/csrc/clr_grad_core.h: the chunk-parallel forward-mode gradient with the
// GPU grid replaced by plain loops (true start states from a sequential replay of the chunks, then per chunk the
// riders and every direction group, then the walk over the chunks per direction), so the algebra can be checked
// against oracle/grad.py in the CPU-only test run.  Not linked into libcelerite_hip.so.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../celerite_amd/csrc/clr_grad_core.h"

using namespace clr;

// slots per chunk for the stored states of the reverse-mode runs (<= 0: as many as steps)
static int g_slot_limit = 0;
extern "C" void hostcheck_grad_set_slot_limit(int n) { g_slot_limit = n; }
// stored states at least this many steps apart, the ones in between rebuilt forwards by the sweep (GradStore::span)
static int g_span = 1;
extern "C" void hostcheck_grad_set_span(int n) { g_span = n > 1 ? n : 1; }

template <int JR, int JC>
static int run_grad(int N, int nchunk, double jitter, const double* a_real, const double* c_real, const double* a_comp,
                    const double* b_comp, const double* c_comp, const double* d_comp, const double* t,
                    const double* diag, const double* y, double* logdet, double* quad, double* grad) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  constexpr int J = Wd::J;
  const int L = (N + nchunk - 1) / nchunk;
  nchunk = (N + L - 1) / L;
  Problem<JR, JC> p;
  p.load(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter);
  std::vector<double> starts((size_t)nchunk * Wd::START, 0.0), riders((size_t)nchunk * Sh::RID),
      gout((size_t)nchunk * Sh::NG * Sh::OUT, 0.0);
  double ld = 0.0, qd = 0.0;
  int bad = 0;
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
    double l, q, en[Wd::START];
    int fl;
    replay_chunk<JR, JC, 0, true>(p, src, L, N, (int)first, c ? &starts[(size_t)c * Wd::START] : nullptr, &l, &q, &fl,
                                  nullptr, nullptr, nullptr, nullptr, 0, en);
    ld += l; qd += q; bad |= fl;
    if (c + 1 < nchunk) memcpy(&starts[(size_t)(c + 1) * Wd::START], en, sizeof(en));
  }
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    const double* st = c ? &starts[(size_t)c * Wd::START] : nullptr;
    {
      DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
      grad_riders_chunk<JR, JC, true>(p, src, L, N, (int)first, st, &riders[(size_t)c * Sh::RID]);
    }
    for (int g = 0; g < Sh::GROUPS; ++g) {
      int kind, term, q0, q1;
      grad_group<JR, JC>(g, &kind, &term, &q0, &q1);
      DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
      grad_chunk<JR, JC, true>(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter, src, L, N, (int)first, st, g,
                               &gout[((size_t)c * Sh::NG + q0) * Sh::OUT],
                               q1 >= 0 ? &gout[((size_t)c * Sh::NG + q1) * Sh::OUT] : nullptr);
    }
  }
  for (int q = 0; q < Sh::NG; ++q) {
    double dld, dq;
    grad_combine<J>(nchunk, riders.data(), &gout[(size_t)q * Sh::OUT], (long)Sh::NG * Sh::OUT, &dld, &dq);
    grad[q] = -0.5 * (dq + dld);
  }
  *logdet = ld;
  *quad = qd;
  return bad;
}

// The reverse-mode path (grad_riders_chunk with its per-sample record, grad_adjoint_walk, grad_backward_chunk).
// *mismatch: the largest relative difference between the adjoint a chunk's reverse sweep arrives at for its first
// sample and the one the walk over the riders predicted for the end of the previous chunk.
template <int JR, int JC>
static int run_grad_reverse(int N, int nchunk, double jitter, const double* a_real, const double* c_real,
                            const double* a_comp, const double* b_comp, const double* c_comp, const double* d_comp,
                            const double* t, const double* diag, const double* y, double* logdet, double* quad,
                            double* grad, double* mismatch, int K, double* drift, double* stored_fraction) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  constexpr int J = Wd::J, ADJ = Wd::START;
  const int L = (N + nchunk - 1) / nchunk;
  nchunk = (N + L - 1) / L;
  Problem<JR, JC> p;
  p.load(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter);
  std::vector<double> starts((size_t)nchunk * Wd::START, 0.0), ends((size_t)nchunk * Wd::START), riders((size_t)nchunk * Sh::RID),
      rec((size_t)N * (J + 2)), adj((size_t)nchunk * ADJ);
  // stored states: K > 0 every K steps, K == 0 where the accumulated decay reaches the growth budget, K < 0 none
  const int nck = K < 0 ? 0 : L;
  std::vector<double> ck((size_t)nchunk * std::max(nck, 1) * Wd::START, 0.0), counts(nchunk, 0.0);
  std::vector<unsigned char> flags((size_t)nchunk * L, 0);
  auto store_of = [&](int c) {
    GradStore st;
    if (K >= 0) {
      st.ck = &ck[(size_t)c * nck * Wd::START];
      st.flag = &flags[(size_t)c * L];
      st.K = K;
      st.nalloc = g_slot_limit > 0 ? std::min(g_slot_limit, nck) : nck;
      st.count = &counts[c];
      st.span = g_span;
    }
    return st;
  };
  double worst_drift = 0.0;
  double ld = 0.0, qd = 0.0;
  int bad = 0;
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
    double l, q, en[Wd::START];
    int fl;
    replay_chunk<JR, JC, 0, true>(p, src, L, N, (int)first, c ? &starts[(size_t)c * Wd::START] : nullptr, &l, &q, &fl,
                                  nullptr, nullptr, nullptr, nullptr, 0, en);
    ld += l; qd += q; bad |= fl;
    if (c + 1 < nchunk) memcpy(&starts[(size_t)(c + 1) * Wd::START], en, sizeof(en));
  }
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
    grad_riders_chunk<JR, JC, true>(p, src, L, N, (int)first, c ? &starts[(size_t)c * Wd::START] : nullptr,
                                    &riders[(size_t)c * Sh::RID], &rec[(size_t)first * (J + 2)], 1, &ends[(size_t)c * Wd::START],
                                    store_of(c));
  }
  grad_adjoint_walk<J>(nchunk, riders.data(), adj.data());
  std::vector<double> total(Sh::NG, 0.0);
  double worst = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
    double out[Sh::NG], adj0[ADJ], dr = 0.0;
    grad_backward_chunk<JR, JC, true>(p, src, L, N, (int)first, &ends[(size_t)c * Wd::START], &adj[(size_t)c * ADJ],
                                      &rec[(size_t)first * (J + 2)], 1, out, adj0, store_of(c), &dr,
                                      c ? &starts[(size_t)c * Wd::START] : nullptr);
    if (!(dr <= worst_drift)) worst_drift = dr;
    for (int q = 0; q < Sh::NG; ++q) total[q] += out[q];
    if (c > 0) {
      const double* want = &adj[(size_t)(c - 1) * ADJ];
      double mx = 0.0, df = 0.0;
      for (int i = 0; i < ADJ; ++i) { mx = std::fmax(mx, std::fabs(want[i])); df = std::fmax(df, std::fabs(want[i] - adj0[i])); }
      if (mx > 0.0) worst = std::fmax(worst, df / mx);
    }
  }
  for (int q = 0; q < Sh::NG; ++q) grad[q] = -0.5 * total[q];
  *logdet = ld;
  *quad = qd;
  *mismatch = worst;
  *drift = worst_drift;
  if (stored_fraction) {
    double total = 0.0;
    for (double x : counts) total += x;
    *stored_fraction = total / N;
  }
  return bad;
}

// Riders of every chunk two ways: along the base trajectory (grad_riders_chunk) and from the scan's element of the
// chunk + its start state (summarize_chunk -> grad_riders_from_element).  Returns the largest difference relative to
// the largest entry of each of AA, eta, JJ over the chunks.
template <int JR, int JC>
static int run_riders_check(int N, int nchunk, double jitter, const double* a_real, const double* c_real,
                            const double* a_comp, const double* b_comp, const double* c_comp, const double* d_comp,
                            const double* t, const double* diag, const double* y, double* worst /* [3] */) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  constexpr int J = Wd::J;
  const int L = (N + nchunk - 1) / nchunk;
  nchunk = (N + L - 1) / L;
  Problem<JR, JC> p;
  p.load(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter);
  std::vector<double> start(Wd::START, 0.0);
  worst[0] = worst[1] = worst[2] = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    const long first = (long)c * L;
    const double* st = c ? start.data() : nullptr;
    double direct[Sh::RID], fromel[Sh::RID], elem[Wd::ELEM], ld0, q0, en[Wd::START];
    int fl;
    {
      DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
      grad_riders_chunk<JR, JC, true>(p, src, L, N, (int)first, st, direct);
    }
    {
      DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
      summarize_chunk<JR, JC, true>(p, src, L, (int)first, N, true, elem, &ld0, &q0, &fl);
    }
    grad_riders_from_element<J>(elem, st, fromel);
    const int lo[4] = {0, J * J, J * J + J, Sh::RID};
    for (int part = 0; part < 3; ++part) {
      if (part == 0 && c == nchunk - 1) continue;  // (AA of the last chunk is never used)
      double big = 0.0, dev = 0.0;
      for (int i = lo[part]; i < lo[part + 1]; ++i) { big = std::fmax(big, std::fabs(direct[i])); dev = std::fmax(dev, std::fabs(direct[i] - fromel[i])); }
      if (big > 0.0) worst[part] = std::fmax(worst[part], dev / big);
    }
    {
      DirectSeries src{t + first, diag + first, y + first, 1, L, L, (long)N - first};
      double l, q;
      replay_chunk<JR, JC, 0, true>(p, src, L, N, (int)first, st, &l, &q, &fl, nullptr, nullptr, nullptr, nullptr, 0, en);
      memcpy(start.data(), en, sizeof(en));
    }
  }
  return 0;
}

extern "C" int hostcheck_riders(int N, int JR, int JC, int nchunk, double jitter, const double* a_real,
                                const double* c_real, const double* a_comp, const double* b_comp, const double* c_comp,
                                const double* d_comp, const double* t, const double* diag, const double* y, double* worst) {
#define GCASE(R, C) if (JR == R && JC == C) return run_riders_check<R, C>(N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, worst);
  GCASE(1, 0) GCASE(2, 0) GCASE(0, 1) GCASE(1, 1) GCASE(2, 1) GCASE(0, 2) GCASE(2, 3) GCASE(3, 2) GCASE(8, 0) GCASE(0, 4)
#undef GCASE
  return -1;
}

extern "C" int hostcheck_grad_reverse(int N, int JR, int JC, int nchunk, double jitter, const double* a_real,
                                      const double* c_real, const double* a_comp, const double* b_comp,
                                      const double* c_comp, const double* d_comp, const double* t, const double* diag,
                                      const double* y, double* logdet, double* quad, double* grad, double* mismatch,
                                      int K, double* drift, double* stored_fraction) {
#define GCASE(R, C) if (JR == R && JC == C) return run_grad_reverse<R, C>(N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, logdet, quad, grad, mismatch, K, drift, stored_fraction);
  GCASE(1, 0) GCASE(2, 0) GCASE(0, 1) GCASE(1, 1) GCASE(2, 1) GCASE(0, 2) GCASE(2, 3) GCASE(3, 2) GCASE(8, 0) GCASE(0, 4)
#undef GCASE
  return -1;
}

extern "C" int hostcheck_grad(int N, int JR, int JC, int nchunk, double jitter, const double* a_real,
                              const double* c_real, const double* a_comp, const double* b_comp, const double* c_comp,
                              const double* d_comp, const double* t, const double* diag, const double* y,
                              double* logdet, double* quad, double* grad) {
#define GCASE(R, C) if (JR == R && JC == C) return run_grad<R, C>(N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, logdet, quad, grad);
  GCASE(1, 0) GCASE(2, 0) GCASE(0, 1) GCASE(1, 1) GCASE(2, 1) GCASE(0, 2) GCASE(2, 3) GCASE(3, 2) GCASE(8, 0) GCASE(0, 4)
#undef GCASE
  return -1;
}
