// tests/hostcheck/hostcheck_grad.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host (g++) instantiation of celerite_amd/csrc/clr_grad_core.h: the chunk-parallel forward-mode gradient with the
// GPU grid replaced by plain loops (true start states from a sequential replay of the chunks, then per chunk the
// riders and every direction group, then the walk over the chunks per direction), so the algebra can be checked
// against oracle/grad.py in the CPU-only test run.  Not linked into libcelerite_hip.so.
#include <cstring>
#include <vector>

#include "../../celerite_amd/csrc/clr_grad_core.h"

using namespace clr;

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

extern "C" int hostcheck_grad(int N, int JR, int JC, int nchunk, double jitter, const double* a_real,
                              const double* c_real, const double* a_comp, const double* b_comp, const double* c_comp,
                              const double* d_comp, const double* t, const double* diag, const double* y,
                              double* logdet, double* quad, double* grad) {
#define GCASE(R, C) if (JR == R && JC == C) return run_grad<R, C>(N, nchunk, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, logdet, quad, grad);
  GCASE(1, 0) GCASE(2, 0) GCASE(0, 1) GCASE(1, 1) GCASE(2, 1) GCASE(0, 2) GCASE(2, 3) GCASE(3, 2) GCASE(8, 0) GCASE(0, 4)
#undef GCASE
  return -1;
}
