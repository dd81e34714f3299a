/*
 * oracle/celerite_ref_quad.c -- the SAME recurrences as celerite_ref_loops.inc
 * (cholesky.h:126-179 factorisation, :236-260 solve, :343-357 dot_solve), in the
 * same operation order, carried in IEEE binary128 (__float128, libquadmath) from
 * the double-precision inputs.  TEST INFRASTRUCTURE ONLY (see celerite_ref.h).
 *
 * Purpose: at N = 1e5 the device's materialised factor
 * deviates from the double-precision oracle by a few 1e-11 -- this file provides
 * "the truth" (113-bit significand: rounding errors ~1e-34 per operation, i.e.
 * exact to double precision after 1e5 steps at any conditioning these families
 * reach) so that tests can attribute the deviation: scan vs truth, sequential
 * double vs truth.  It is a celerite-terms-only, one-right-hand-side helper, not
 * a second oracle: its results are compared, never shipped.
 *
 * Pinning: tests/test_oracle.py checks it against the double-precision
 * restatement (agreement to the latter's rounding) and against the mpmath dense
 * LDL^T of oracle/dense.py on small cases.
 */
#include <quadmath.h>
#include <stdlib.h>

#include "celerite_ref.h"

typedef __float128 q_t;

/* Factor + log det + K^-1 y + y^T K^-1 y of ONE problem.  Outputs (any may be
 * NULL) are the binary128 values rounded to double: W [J x N] and D [N] in the
 * reference's storage (W[j + J n]), x = solve(y) [N]. */
int refq_factor_solve(double jitter, int J_real, const double* a_real, const double* c_real,
                      int J_comp, const double* a_comp, const double* b_comp,
                      const double* c_comp, const double* d_comp,
                      int N, const double* t, const double* diag, const double* y,
                      double* W_out, double* D_out, double* x_out, double* logdet_out, double* quad_out)
{
  const int J = J_real + 2 * J_comp;
  if (N < 1 || J < 1) return REF_DIMENSION_MISMATCH;
  q_t* phi = (q_t*)malloc(sizeof(q_t) * (size_t)J * (size_t)N);
  q_t* u = (q_t*)malloc(sizeof(q_t) * (size_t)J * (size_t)N);
  q_t* W = (q_t*)malloc(sizeof(q_t) * (size_t)J * (size_t)N);
  q_t* D = (q_t*)malloc(sizeof(q_t) * (size_t)N);
  q_t* S = (q_t*)calloc((size_t)J * (size_t)J, sizeof(q_t));
  q_t* f = (q_t*)malloc(sizeof(q_t) * (size_t)J);
  q_t* x = (q_t*)malloc(sizeof(q_t) * (size_t)N);
  int status = REF_OK;
  if (!phi || !u || !W || !D || !S || !f || !x) { status = REF_DIMENSION_MISMATCH; goto done; }

  /* cholesky.h:100-117: D_0 = diag_0 + jitter + sum a, W_0 = (1 | cos, sin) / D_0 */
  {
    q_t a_sum = (q_t)jitter;
    for (int j = 0; j < J_real; ++j) a_sum += (q_t)a_real[j];
    for (int j = 0; j < J_comp; ++j) a_sum += (q_t)a_comp[j];
    for (int n = 0; n < N; ++n) D[n] = (q_t)diag[n] + a_sum;
    const q_t t0 = (q_t)t[0];
    for (int j = 0; j < J_real; ++j) W[j] = 1.0Q / D[0];
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const q_t d = (q_t)d_comp[j] * t0;
      W[k] = cosq(d) / D[0];
      W[k + 1] = sinq(d) / D[0];
    }
  }
  q_t Dn = D[0];
  for (int n = 1; n < N; ++n) {                             /* :126-179 */
    q_t* phin = phi + (size_t)J * (n - 1);
    q_t* un = u + (size_t)J * (n - 1);
    q_t* Wn = W + (size_t)J * n;
    const q_t* Wp = W + (size_t)J * (n - 1);
    const q_t tn = (q_t)t[n], dx = tn - (q_t)t[n - 1];
    for (int j = 0; j < J_real; ++j) {
      phin[j] = expq(-(q_t)c_real[j] * dx);
      un[j] = (q_t)a_real[j];
      Wn[j] = 1.0Q;
    }
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const q_t a = (q_t)a_comp[j], b = (q_t)b_comp[j], d = (q_t)d_comp[j] * tn;
      const q_t cd = cosq(d), sd = sinq(d);
      const q_t value = expq(-(q_t)c_comp[j] * dx);
      phin[k] = value;
      phin[k + 1] = value;
      un[k] = a * cd + b * sd;
      un[k + 1] = a * sd - b * cd;
      Wn[k] = cd;
      Wn[k + 1] = sd;
    }
    for (int j = 0; j < J; ++j) {                            /* :154-160 */
      const q_t phij = phin[j], xj = Dn * Wp[j];
      for (int k = 0; k <= j; ++k)
        S[k + J * j] = phij * (phin[k] * (S[k + J * j] + xj * Wp[k]));
    }
    Dn = D[n];
    for (int j = 0; j < J; ++j) {                            /* :163-175 */
      const q_t uj = un[j];
      q_t xj = Wn[j];
      for (int k = 0; k < j; ++k) {
        const q_t tmp = un[k] * S[k + J * j];
        Dn -= 2.0Q * (uj * tmp);
        xj -= tmp;
        Wn[k] -= uj * S[k + J * j];
      }
      const q_t tmp = uj * S[j + J * j];
      Dn -= uj * tmp;
      Wn[j] = xj - tmp;
    }
    if (Dn < 0) { status = REF_LINALG; goto done; }          /* :176 */
    D[n] = Dn;
    for (int j = 0; j < J; ++j) Wn[j] /= Dn;
  }
  {
    q_t ld = 0.0Q;                                           /* solver.h:74-81 / cholesky.h:208 */
    for (int n = 0; n < N; ++n) ld += logq(D[n]);
    if (logdet_out) *logdet_out = (double)ld;
  }
  /* forward sweep (:240-248), quadratic form (:343-357), / D (:249), backward sweep (:252-259) */
  for (int j = 0; j < J; ++j) f[j] = 0.0Q;
  x[0] = (q_t)y[0];
  for (int n = 1; n < N; ++n) {
    const q_t* phin = phi + (size_t)J * (n - 1);
    const q_t* un = u + (size_t)J * (n - 1);
    const q_t* Wp = W + (size_t)J * (n - 1);
    const q_t xnm1 = x[n - 1];
    q_t acc = (q_t)y[n];
    for (int j = 0; j < J; ++j) {
      const q_t value = phin[j] * (f[j] + Wp[j] * xnm1);
      f[j] = value;
      acc -= un[j] * value;
    }
    x[n] = acc;
  }
  {
    q_t quad = 0.0Q;
    for (int n = 0; n < N; ++n) quad += x[n] * x[n] / D[n];
    if (quad_out) *quad_out = (double)quad;
  }
  for (int n = 0; n < N; ++n) x[n] /= D[n];
  for (int j = 0; j < J; ++j) f[j] = 0.0Q;
  for (int n = N - 2; n >= 0; --n) {
    const q_t* phin = phi + (size_t)J * n;
    const q_t* un = u + (size_t)J * n;
    const q_t* Wn = W + (size_t)J * n;
    const q_t xnp1 = x[n + 1];
    q_t acc = x[n];
    for (int j = 0; j < J; ++j) {
      const q_t value = phin[j] * (f[j] + un[j] * xnp1);
      f[j] = value;
      acc -= Wn[j] * value;
    }
    x[n] = acc;
  }
  if (W_out) for (size_t i = 0; i < (size_t)J * (size_t)N; ++i) W_out[i] = (double)W[i];
  if (D_out) for (int n = 0; n < N; ++n) D_out[n] = (double)D[n];
  if (x_out) for (int n = 0; n < N; ++n) x_out[n] = (double)x[n];
done:
  free(phi); free(u); free(W); free(D); free(S); free(f); free(x);
  return status;
}
