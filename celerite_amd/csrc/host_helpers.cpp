// celerite_amd/csrc/host_helpers.cpp
//
// O(J) scalar helpers the Python layer imports from the compiled module
// (celerite/terms.py:18 needs get_kernel_value, get_psd_value,
// check_coefficients).  Plain host C++: no device work, not on the hot path
// (SURVEY.md section 2, rows 5-6).  Semantics follow
// cpp/include/celerite/utils.h:27-163 and poly.h (Sturm's theorem on the
// numerator polynomial of the PSD, in the variable omega^2).
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/celerite_hip.h"

namespace {

typedef std::vector<double> Poly;  // coefficients, highest power first
const double kPolyTol = 1e-10;     // poly.h:11

double eval(const Poly& p, double x) {
  double acc = 0.0;
  for (double c : p) acc = acc * x + c;
  return acc;
}

Poly add(const Poly& p, const Poly& q) {
  const size_t n = std::max(p.size(), q.size());
  Poly out(n, 0.0);
  for (size_t i = 0; i < p.size(); ++i) out[n - p.size() + i] += p[i];
  for (size_t i = 0; i < q.size(); ++i) out[n - q.size() + i] += q[i];
  return out;
}

Poly mul(const Poly& p, const Poly& q) {
  Poly out(p.size() + q.size() - 1, 0.0);
  for (size_t i = 0; i < p.size(); ++i)
    for (size_t j = 0; j < q.size(); ++j) out[i + j] += p[i] * q[j];
  return out;
}

// Remainder of u / v with leading near-zeros (|c| < tol) stripped (poly.h:49-70).
Poly rem(const Poly& u, const Poly& v) {
  const int m = (int)u.size() - 1, n = (int)v.size() - 1, steps = m - n + 1;
  Poly r = u;
  const double scale = 1.0 / v[0];
  for (int k = 0; k < steps; ++k) {
    const double d = scale * r[k];
    for (int i = 0; i <= n; ++i) r[k + i] -= d * v[i];
  }
  int first = 0;
  while (first < m && std::fabs(r[first]) < kPolyTol) ++first;
  return Poly(r.begin() + first, r.end());
}

Poly derivative(const Poly& p) {
  const int n = (int)p.size() - 1;
  Poly d(p.begin(), p.begin() + n);
  for (int i = 0; i < n; ++i) d[i] *= (n - i);
  return d;
}

int sign(double v) { return (0.0 < v) - (v < 0.0); }

// Number of roots in (0, inf) via sign changes of the Sturm chain at 0 and at
// +inf (poly.h:106-138).
int count_positive_roots(const Poly& p) {
  if (p.size() <= 1) return 0;
  const int n = (int)p.size() - 1;
  Poly p0 = p, p1 = derivative(p0);
  int at0 = sign(p1.back()), atinf = sign(p1.front());
  int count = (sign(p0.back()) != at0) - (sign(p0.front()) != atinf);
  for (int k = 0; k < n; ++k) {
    Poly next = rem(p0, p1);
    for (double& c : next) c *= -1.0;
    p0 = p1;
    p1 = next;
    const int s0 = sign(p1.back()), sinf = sign(p1.front());
    count += (at0 != s0);
    count -= (atinf != sinf);
    at0 = s0;
    atinf = sinf;
    if (p1.size() == 1) break;
  }
  return count;
}

}  // namespace

extern "C" {

double clr_kernel_value(int J_real, const double* a_real, const double* c_real, int J_comp,
                        const double* a_comp, const double* b_comp, const double* c_comp,
                        const double* d_comp, double tau) {
  const double t = std::fabs(tau);  // utils.h:106-132
  double k = 0.0;
  for (int i = 0; i < J_real; ++i) k += a_real[i] * std::exp(-c_real[i] * t);
  for (int i = 0; i < J_comp; ++i)
    k += std::exp(-c_comp[i] * t) *
         (a_comp[i] * std::cos(d_comp[i] * t) + b_comp[i] * std::sin(d_comp[i] * t));
  return k;
}

double clr_psd_value(int J_real, const double* a_real, const double* c_real, int J_comp,
                     const double* a_comp, const double* b_comp, const double* c_comp,
                     const double* d_comp, double omega) {
  const double w2 = omega * omega;  // utils.h:134-163
  double p = 0.0;
  for (int i = 0; i < J_real; ++i) {
    const double a = a_real[i], c = c_real[i];
    p += a * c / (c * c + w2);
  }
  for (int i = 0; i < J_comp; ++i) {
    const double a = a_comp[i], b = b_comp[i], c = c_comp[i], d = d_comp[i];
    const double w02 = c * c + d * d;
    p += ((a * c + b * d) * w02 + (a * c - b * d) * w2) /
         (w2 * w2 + 2.0 * (c * c - d * d) * w2 + w02 * w02);
  }
  return std::sqrt(2.0 / M_PI) * p;
}

int clr_check_coefficients(int n_a_real, const double* a_real, int n_c_real,
                           const double* c_real, int n_a_comp, const double* a_comp,
                           int n_b_comp, const double* b_comp, int n_c_comp,
                           const double* c_comp, int n_d_comp, const double* d_comp) {
  if (n_a_real != n_c_real) return 0;  // utils.h:41-44
  if (n_a_comp != n_b_comp || n_a_comp != n_c_comp || n_a_comp != n_d_comp) return 0;

  // PSD of every term as num_k(w^2) / den_k(w^2) (utils.h:46-83)
  std::vector<Poly> num, den;
  for (int i = 0; i < n_a_real; ++i) {
    const double a = a_real[i], c = c_real[i], c2 = c * c;
    num.push_back(Poly{a * c, a * c * c2});
    den.push_back(Poly{1.0, 2.0 * c2, c2 * c2});
  }
  for (int i = 0; i < n_a_comp; ++i) {
    const double a = a_comp[i], b = b_comp[i], c = c_comp[i], d = d_comp[i];
    const double c2 = c * c, d2 = d * d, w0 = c2 + d2;
    num.push_back(Poly{a * c - b * d, (a * c + b * d) * w0});
    den.push_back(Poly{1.0, 2.0 * (c2 - d2), w0 * w0});
  }

  // numerator over the common denominator (utils.h:85-94)
  const int n = (int)num.size();
  Poly total(1, 0.0);
  for (int i = 0; i < n; ++i) {
    Poly term = num[i];
    for (int j = 0; j < n; ++j)
      if (j != i) term = mul(term, den[j]);
    total = add(total, term);
  }
  while (total.size() > 1 && std::fabs(total[0]) < kPolyTol) total.erase(total.begin());  // :97-98
  if (eval(total, 0.0) < 0.0) return 0;                                                   // :100
  return count_positive_roots(total) == 0;                                                // :103-104
}

}  // extern "C"
