// celerite_amd/csrc/carma.hip -- the reference's comparison solver `CARMASolver` (cpp/include/celerite/carma.h,
// bound at celerite/solver.cpp:200-235; SURVEY.md 8(f) row 4, used by tests/test_celerite.py:22-42).
//
// Two parts:
//   * model set-up on the host (O(p^3), p = autoregressive order <= 32): roots of the AR / MA polynomials from
//     carma_pack's parameters, the rotation into the diagonalised state space (observation row b, stationary
//     covariance V), and the conversion to celerite coefficients.  Parameter algebra like terms.py: no device work.
//   * the Kalman-filter log-likelihood on the device: the filter is sequential in n (each step needs the state
//     the previous one left), so ONE wave walks the series with the state in LDS -- P (p x p complex), V, b,
//     the gain K, the propagators lambda -- lane r owning row r for the gain / mean, and the p^2 covariance
//     entries dealt round the lanes for the rank-one update and the propagation.  ~1 us per sample; the
//     reference's test runs N = 100.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <string>
#include <vector>

#include "../../include/celerite_hip.h"
#include "clr_carma.h"
#include "clr_wide.h"

namespace clr {

namespace {

typedef std::complex<double> cplx;

// carma.h:15-29: consecutive (log c, log b) pairs are quadratic factors x^2 + b x + c, an odd last one a real root
std::vector<cplx> roots_of(const double* par, int n) {
  std::vector<cplx> r((size_t)n);
  if (n % 2 == 1) r[(size_t)n - 1] = -std::exp(par[n - 1]);
  for (int i = 0; i + 1 < n; i += 2) {
    const cplx b = std::exp(par[i + 1]), c = std::exp(par[i]);
    const cplx disc = std::sqrt(b * b - 4.0 * c);
    r[(size_t)i] = 0.5 * (-b + disc);
    r[(size_t)i + 1] = 0.5 * (-b - disc);
  }
  return r;
}

// carma.h:31-44: monic polynomial with the given roots, lowest power first
std::vector<cplx> poly_of(const std::vector<cplx>& r) {
  const int n = (int)r.size() + 1;
  std::vector<cplx> c((size_t)n, cplx(0.0));
  if (n == 1) { c[0] = 1.0; return c; }
  c[0] = -r[0];
  c[1] = 1.0;
  for (int i = 1; i < n - 1; ++i) {
    for (int j = n - 1; j >= 1; --j) c[(size_t)j] = c[(size_t)j - 1] - r[(size_t)i] * c[(size_t)j];
    c[0] *= -r[(size_t)i];
  }
  return c;
}

// U x = rhs by Gaussian elimination with full pivoting (the reference uses Eigen::FullPivLU, carma.h:157)
bool solve_full_pivot(std::vector<cplx> U, std::vector<cplx> rhs, int p, std::vector<cplx>& x) {
  std::vector<int> colperm((size_t)p);
  for (int i = 0; i < p; ++i) colperm[(size_t)i] = i;
  for (int k = 0; k < p; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int i = k; i < p; ++i)
      for (int j = k; j < p; ++j) {
        const double m = std::abs(U[(size_t)i * p + j]);
        if (m > best) { best = m; pr = i; pc = j; }
      }
    if (!(best > 0.0)) return false;
    if (pr != k) {
      for (int j = 0; j < p; ++j) std::swap(U[(size_t)pr * p + j], U[(size_t)k * p + j]);
      std::swap(rhs[(size_t)pr], rhs[(size_t)k]);
    }
    if (pc != k) {
      for (int i = 0; i < p; ++i) std::swap(U[(size_t)i * p + pc], U[(size_t)i * p + k]);
      std::swap(colperm[(size_t)pc], colperm[(size_t)k]);
    }
    for (int i = k + 1; i < p; ++i) {
      const cplx f = U[(size_t)i * p + k] / U[(size_t)k * p + k];
      for (int j = k; j < p; ++j) U[(size_t)i * p + j] -= f * U[(size_t)k * p + j];
      rhs[(size_t)i] -= f * rhs[(size_t)k];
    }
  }
  std::vector<cplx> z((size_t)p);
  for (int i = p - 1; i >= 0; --i) {
    cplx acc = rhs[(size_t)i];
    for (int j = i + 1; j < p; ++j) acc -= U[(size_t)i * p + j] * z[(size_t)j];
    z[(size_t)i] = acc / U[(size_t)i * p + i];
  }
  x.assign((size_t)p, cplx(0.0));
  for (int i = 0; i < p; ++i) x[(size_t)colperm[(size_t)i]] = z[(size_t)i];
  return true;
}

inline bool close6(double a, double b) { return std::abs(a - b) <= 1e-6; }                    // utils.h:16-20
inline cplx lse(const cplx& a, const cplx& b) { return b + std::log(cplx(1.0) + std::exp(a - b)); }  // utils.h:22-25

}  // namespace

int carma_setup(double log_sigma, int p, const double* ar, int q, const double* ma, CarmaModel& M, std::string& err) {
  if (q >= p) { err = "dimension mismatch"; return CLR_DIMENSION_MISMATCH; }  // carma.h:59
  if (p > CLR_CARMA_MAX_ORDER) { err = "CARMA order above CLR_CARMA_MAX_ORDER"; return CLR_UNSUPPORTED; }
  M.p = p; M.q = q; M.sigma = std::exp(log_sigma);
  M.arroots = roots_of(ar, p);
  const std::vector<cplx> maroots = roots_of(ma, q);
  M.beta = poly_of(maroots);
  const cplx b0 = M.beta[0];
  for (cplx& v : M.beta) v /= b0;  // carma.h:69
  // rotation into the diagonalised space (carma.h:141-165): U_ij = r_j^i, b = [beta 0..] U, J = U \ (sigma e_p),
  // V_ij = -J_i conj(J_j) / (r_i + conj(r_j))
  std::vector<cplx> U((size_t)p * p);
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < p; ++j) U[(size_t)i * p + j] = std::pow(M.arroots[(size_t)j], i);
  M.b.assign((size_t)p, cplx(0.0));
  for (int j = 0; j < p; ++j)
    for (int i = 0; i <= q; ++i) M.b[(size_t)j] += M.beta[(size_t)i] * U[(size_t)i * p + j];
  std::vector<cplx> e((size_t)p, cplx(0.0)), Jv;
  e[(size_t)p - 1] = M.sigma;
  if (!solve_full_pivot(U, e, p, Jv)) { err = "CARMA model: singular rotation (repeated autoregressive roots)"; return CLR_INVALID_ARGUMENT; }
  M.V.assign((size_t)p * p, cplx(0.0));
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < p; ++j)
      M.V[(size_t)i * p + j] = -Jv[(size_t)i] * std::conj(Jv[(size_t)j]) / (M.arroots[(size_t)i] + std::conj(M.arroots[(size_t)j]));
  // advance_time raises lambda_base = exp(r) to the power dt (carma.h:61-62,208): std::pow of a complex base
  // goes through log(lambda_base), whose imaginary part is the PRINCIPAL argument -- not Im r when |Im r| > pi.
  // A positive real base takes the real pow.  Either way lambda(dt) = exp(dt * loglam) with:
  M.loglam.resize((size_t)p);
  for (int i = 0; i < p; ++i) {
    const cplx lb = std::exp(M.arroots[(size_t)i]);
    M.loglam[(size_t)i] = (lb.imag() == 0.0 && lb.real() > 0.0) ? cplx(std::log(lb.real()), 0.0) : std::log(lb);
  }
  return CLR_OK;
}

// carma.h:74-139
void carma_celerite_coeffs(const CarmaModel& M, std::vector<double> out[6]) {
  const int p = M.p, q = M.q;
  for (int i = 0; i < 6; ++i) out[i].clear();
  std::vector<double>&ar = out[0], &cr = out[1], &a = out[2], &b = out[3], &c = out[4], &d = out[5];
  for (int k = 0; k < p; ++k) {
    const cplx rk = M.arroots[(size_t)k];
    cplx t1 = std::log(M.beta[0]), t2 = t1;
    for (int l = 1; l <= q; ++l) {
      t1 = lse(t1, std::log(M.beta[(size_t)l]) + cplx((double)l) * std::log(rk));
      t2 = lse(t2, std::log(M.beta[(size_t)l]) + cplx((double)l) * std::log(-rk));
    }
    cplx full = 2.0 * std::log(M.sigma) + t1 + t2 - std::log(cplx(-rk.real()));
    for (int l = 0; l < p; ++l)
      if (l != k) full -= std::log(M.arroots[(size_t)l] - rk) + std::log(std::conj(M.arroots[(size_t)l]) + rk);
    full = std::exp(full);
    if (close6(full.imag(), 0.0) && close6(rk.imag(), 0.0)) {
      ar.push_back(0.5 * full.real());
      cr.push_back(-rk.real());
      continue;
    }
    bool seen = false;  // the conjugate partner of a pair already recorded
    for (size_t l = 0; l < a.size() && !seen; ++l)
      seen = close6(a[l], full.real()) && close6(b[l], -full.imag()) && close6(c[l], -rk.real()) && close6(d[l], rk.imag());
    if (!seen) {
      a.push_back(full.real());
      b.push_back(full.imag());
      c.push_back(-rk.real());
      d.push_back(-rk.imag());
    }
  }
}

namespace {

constexpr int MAXP = CLR_CARMA_MAX_ORDER;

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmul_conj(double2 a, double2 b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ void wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// model: [b (p) | V (p*p, row-major) | loglam (p)] as (re, im) pairs.  out[0] = log-likelihood; status[0] = 1 when
// a predicted variance went negative (carma_exception, carma.h:185-186).
__global__ void __launch_bounds__(64) carma_filter_kernel(int n, int p, const double2* model, const double* t,
                                                          const double* y, const double* yerr, double* out,
                                                          int* status) {
  __shared__ double2 Pm[MAXP * MAXP], Vm[MAXP * MAXP], bm[MAXP], Km[MAXP], lm[MAXP], xm[MAXP], ll[MAXP];
  const int lane = threadIdx.x, pp = p * p;
  for (int r = lane; r < p; r += 64) { bm[r] = model[r]; ll[r] = model[p + pp + r]; xm[r] = make_double2(0.0, 0.0); }
  for (int e = lane; e < pp; e += 64) { Vm[e] = model[p + e]; Pm[e] = Vm[e]; }  // reset, carma.h:167-173
  wave_fence();
  double total = (double)n * 1.8378770664093453;  // n log(2 pi), carma.h:224
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    // predict (carma.h:175-187): row r of P conj(b), then the two real sums over the rows
    double e_part = 0.0, v_part = 0.0;
    double2 kr = make_double2(0.0, 0.0);
    if (lane < p) {
      for (int s = 0; s < p; ++s) {
        const double2 c = cmul_conj(Pm[lane * p + s], bm[s]);
        kr.x += c.x; kr.y += c.y;
      }
      const double2 bx = cmul(bm[lane], xm[lane]);
      e_part = bx.x;
      v_part = cmul(bm[lane], kr).x;
    }
    const double expectation = row_sum<1>(e_part);
    const double ye = yerr[i];
    const double variance = ye * ye + row_sum<1>(v_part);
    if (variance < 0.0) { bad = 1; break; }  // (wave-uniform)
    // update_state (carma.h:189-202)
    const double resid = y[i] - expectation;
    if (lane < p) {
      kr.x /= variance; kr.y /= variance;
      Km[lane] = kr;
      xm[lane].x += resid * kr.x;
      xm[lane].y += resid * kr.y;
    }
    const bool adv = i < n - 1;
    if (adv && lane < p) {  // advance_time, carma.h:204-211
      const double dt = t[i + 1] - t[i];
      const double mag = exp(dt * ll[lane].x);
      double sn, cs;
      sincos(dt * ll[lane].y, &sn, &cs);
      const double2 lam = make_double2(mag * cs, mag * sn);
      lm[lane] = lam;
      xm[lane] = cmul(xm[lane], lam);
    }
    wave_fence();
    for (int e = lane; e < pp; e += 64) {
      const int r = e / p, s = e - r * p;
      const double2 kk = cmul_conj(Km[r], Km[s]);
      double2 pe = Pm[e];
      pe.x -= variance * kk.x;
      pe.y -= variance * kk.y;
      if (adv) {  // carma.h:212-218: P = V + lam_r (P - V) conj(lam_s)
        const double2 v = Vm[e];
        const double2 dv = make_double2(pe.x - v.x, pe.y - v.y);
        const double2 w = cmul_conj(cmul(lm[r], dv), lm[s]);
        pe = make_double2(v.x + w.x, v.y + w.y);
      }
      Pm[e] = pe;
    }
    wave_fence();
    total += resid * resid / variance + log(variance);  // carma.h:234-235
  }
  if (lane == 0) {
    out[0] = -0.5 * total;
    status[0] = bad;
  }
}

}  // namespace

void launch_carma_filter(int n, int p, const double* model, const double* t, const double* y, const double* yerr,
                         double* out, int* status, hipStream_t s) {
  hipLaunchKernelGGL(carma_filter_kernel, dim3(1), dim3(64), 0, s, n, p, reinterpret_cast<const double2*>(model), t, y,
                     yerr, out, status);
}

}  // namespace clr
