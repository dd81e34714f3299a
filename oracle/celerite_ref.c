/*
 * oracle/celerite_ref.c -- CPU restatement of celerite's CholeskySolver.
 * TEST INFRASTRUCTURE ONLY; see celerite_ref.h for scope, citations and how
 * this oracle is pinned.  Plain C99 + libm (+ pthreads for the optional
 * all-cores batch baseline).
 */
#define _GNU_SOURCE
#include "celerite_ref.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- per-width instantiations of the hot loops -------------------------- */
#define JW 1
#define SUFFIX w1
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 2
#define SUFFIX w2
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 3
#define SUFFIX w3
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 4
#define SUFFIX w4
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 5
#define SUFFIX w5
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 6
#define SUFFIX w6
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 7
#define SUFFIX w7
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 8
#define SUFFIX w8
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 9
#define SUFFIX w9
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 10
#define SUFFIX w10
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 11
#define SUFFIX w11
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 12
#define SUFFIX w12
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 13
#define SUFFIX w13
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 14
#define SUFFIX w14
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 15
#define SUFFIX w15
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW 16
#define SUFFIX w16
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX
#define JW J
#define SUFFIX dyn
#include "celerite_ref_loops.inc"
#undef JW
#undef SUFFIX

#define DISPATCH(J, CALL_FIXED, CALL_DYN) \
  switch (J) {                            \
    case 1: CALL_FIXED(w1); break;        \
    case 2: CALL_FIXED(w2); break;        \
    case 3: CALL_FIXED(w3); break;        \
    case 4: CALL_FIXED(w4); break;        \
    case 5: CALL_FIXED(w5); break;        \
    case 6: CALL_FIXED(w6); break;        \
    case 7: CALL_FIXED(w7); break;        \
    case 8: CALL_FIXED(w8); break;        \
    case 9: CALL_FIXED(w9); break;        \
    case 10: CALL_FIXED(w10); break;      \
    case 11: CALL_FIXED(w11); break;      \
    case 12: CALL_FIXED(w12); break;      \
    case 13: CALL_FIXED(w13); break;      \
    case 14: CALL_FIXED(w14); break;      \
    case 15: CALL_FIXED(w15); break;      \
    case 16: CALL_FIXED(w16); break;      \
    default: CALL_DYN; break;             \
  }

/* ---- object lifetime ----------------------------------------------------- */
static double* dup_vec(const double* src, int n) {
  double* p = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(p, src, sizeof(double) * (size_t)n);
  return p;
}

static void release_arrays(ref_solver* s) {
  free(s->phi); free(s->u); free(s->W); free(s->D);
  free(s->a_real); free(s->c_real); free(s->a_comp); free(s->b_comp);
  free(s->c_comp); free(s->d_comp); free(s->t);
  s->phi = s->u = s->W = s->D = NULL;
  s->a_real = s->c_real = s->a_comp = s->b_comp = s->c_comp = s->d_comp = s->t = NULL;
}

ref_solver* ref_create(void) {
  ref_solver* s = (ref_solver*)calloc(1, sizeof(ref_solver));
  return s; /* computed_ = false: solver.h:23 */
}

void ref_destroy(ref_solver* s) {
  if (!s) return;
  release_arrays(s);
  free(s);
}

/* ---- compute: cholesky.h:41-210 ------------------------------------------ */
int ref_compute(ref_solver* s, double jitter,
                int n_a_real, const double* a_real,
                int n_c_real, const double* c_real,
                int n_a_comp, const double* a_comp,
                int n_b_comp, const double* b_comp,
                int n_c_comp, const double* c_comp,
                int n_d_comp, const double* d_comp,
                int n_A, const double* A,
                int U_rows, int U_cols, const double* U,
                int V_rows, int V_cols, const double* V,
                int n_x, const double* x,
                int n_diag, const double* diag)
{
  const int N = n_x;
  s->computed = 0;                                        /* :57 */

  if (N != n_diag) return REF_DIMENSION_MISMATCH;         /* :59-63 */
  if (n_a_real != n_c_real) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_b_comp) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_c_comp) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_d_comp) return REF_DIMENSION_MISMATCH;

  const int has_general = (n_A != 0);                     /* :65-69 */
  if (has_general && n_A != N) return REF_DIMENSION_MISMATCH;
  if (has_general && U_cols != N) return REF_DIMENSION_MISMATCH;
  if (has_general && V_cols != N) return REF_DIMENSION_MISMATCH;
  if (U_rows != V_rows) return REF_DIMENSION_MISMATCH;

  const int J_general = U_rows, J_real = n_a_real, J_comp = n_a_comp;
  const int J = J_real + 2 * J_comp + J_general;          /* :71-74 */

  release_arrays(s);
  s->N = N;
  s->J = J;
  const long Nm1 = N > 0 ? N - 1 : 0;
  s->phi = (double*)malloc(sizeof(double) * (size_t)(J * Nm1 + 1)); /* :76-78 */
  s->u = (double*)malloc(sizeof(double) * (size_t)(J * Nm1 + 1));
  s->W = (double*)malloc(sizeof(double) * (size_t)((long)J * N + 1));
  s->D = (double*)malloc(sizeof(double) * (size_t)(N + 1));

  s->J_real = J_real;                                     /* :80-87 */
  s->J_comp = J_comp;
  s->a_real = dup_vec(a_real, J_real);
  s->c_real = dup_vec(c_real, J_real);
  s->a_comp = dup_vec(a_comp, J_comp);
  s->b_comp = dup_vec(b_comp, J_comp);
  s->c_comp = dup_vec(c_comp, J_comp);
  s->d_comp = dup_vec(d_comp, J_comp);
  s->t = dup_vec(x, N);

  if (J == 0) {                                           /* :90-95 */
    double ld = 0.0;
    for (int n = 0; n < N; ++n) {
      s->D[n] = diag[n] + jitter;
      ld += log(s->D[n]);
    }
    s->log_det = ld;
    s->computed = 1;
    return REF_OK;
  }

  double sum_ar = 0.0, sum_ac = 0.0;                      /* :98-99 */
  for (int j = 0; j < J_real; ++j) sum_ar += a_real[j];
  for (int j = 0; j < J_comp; ++j) sum_ac += a_comp[j];
  for (int n = 0; n < N; ++n) {
    s->D[n] = ((diag[n] + sum_ar) + sum_ac) + jitter;
    if (has_general) s->D[n] += A[n];
  }

  {                                                       /* :103-117 */
    const double value = 1.0 / s->D[0], t0 = x[0];
    for (int j = 0; j < J_real; ++j) s->W[j] = value;
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const double d = d_comp[j] * t0;
      s->W[k] = cos(d) * value;
      s->W[k + 1] = sin(d) * value;
    }
    for (int j = 0, k = J_real + 2 * J_comp; j < J_general; ++j, ++k)
      s->W[k] = V[(long)j * N + 0] * value;
  }

  double* S = (double*)malloc(sizeof(double) * (size_t)(J * J));
  int status = REF_OK;
#define CALL_FIXED(sfx)                                                        \
  status = factor_loop_##sfx(N, J, J_real, J_comp, J_general, a_real, c_real,  \
                             a_comp, b_comp, c_comp, d_comp, U, V, x, s->phi,  \
                             s->u, s->W, s->D, S)
  DISPATCH(J, CALL_FIXED, CALL_FIXED(dyn))                /* :181-204 */
#undef CALL_FIXED
  free(S);
  if (status != REF_OK) return status;                    /* throw at :176 */

  double ld = 0.0;                                        /* :208 */
  for (int n = 0; n < N; ++n) ld += log(s->D[n]);
  s->log_det = ld;
  s->computed = 1;
  return REF_OK;
}

/* ---- getters: solver.h:74-81 ---------------------------------------------- */
int ref_log_determinant(const ref_solver* s, double* out) {
  if (!s->computed) return REF_NOT_COMPUTED;
  *out = s->log_det;
  return REF_OK;
}

int ref_computed(const ref_solver* s) { return s->computed; }

/* ---- dot_solve: cholesky.h:326-401 ----------------------------------------- */
int ref_dot_solve(const ref_solver* s, int b_rows, const double* b, double* out) {
  if (b_rows != s->N) return REF_DIMENSION_MISMATCH;      /* :327 */
  if (!s->computed) return REF_NOT_COMPUTED;              /* :328 */
  const int N = s->N, J = s->J;
  if (J == 0) {                                           /* :334-339 */
    double r = 0.0;
    for (int n = 0; n < N; ++n) r += b[n] * (b[n] / s->D[n]);
    *out = r;
    return REF_OK;
  }
  double* f = (double*)malloc(sizeof(double) * (size_t)J);
  double r = 0.0;
#define CALL_FIXED(sfx) \
  r = dot_solve_loop_##sfx(N, J, s->phi, s->u, s->W, s->D, b, f)
  DISPATCH(J, CALL_FIXED, CALL_FIXED(dyn))
#undef CALL_FIXED
  free(f);
  *out = r;
  return REF_OK;
}

/* ---- solve: cholesky.h:218-318 --------------------------------------------- */
int ref_solve(const ref_solver* s, int b_rows, int nrhs, const double* b, double* x) {
  if (b_rows != s->N) return REF_DIMENSION_MISMATCH;      /* :219 */
  if (!s->computed) return REF_NOT_COMPUTED;              /* :220 */
  const int N = s->N, J = s->J;
  if (J == 0) {                                           /* :226-231 */
    for (int k = 0; k < nrhs; ++k)
      for (int n = 0; n < N; ++n) x[(long)k * N + n] = b[(long)k * N + n] / s->D[n];
    return REF_OK;
  }
  double* f = (double*)malloc(sizeof(double) * (size_t)J);
  for (int k = 0; k < nrhs; ++k) {
    const double* bk = b + (long)k * N;
    double* xk = x + (long)k * N;
#define CALL_FIXED(sfx) solve_loop_##sfx(N, J, s->phi, s->u, s->W, s->D, bk, xk, f)
    DISPATCH(J, CALL_FIXED, CALL_FIXED(dyn))
#undef CALL_FIXED
  }
  free(f);
  return REF_OK;
}

/* ---- dot_L: cholesky.h:409-431 --------------------------------------------- */
int ref_dot_L(const ref_solver* s, int z_rows, int nrhs, const double* z, double* y) {
  if (z_rows != s->N) return REF_DIMENSION_MISMATCH;      /* :410 */
  if (!s->computed) return REF_NOT_COMPUTED;              /* :411 */
  const int N = s->N, J = s->J;
  double* f = (double*)malloc(sizeof(double) * (size_t)(J > 0 ? J : 1));
  for (int k = 0; k < nrhs; ++k) {
    const double* zk = z + (long)k * N;
    double* yk = y + (long)k * N;
    for (int j = 0; j < J; ++j) f[j] = 0.0;
    double tmp = zk[0] * sqrt(s->D[0]);                   /* :421-422 */
    yk[0] = tmp;
    for (int n = 1; n < N; ++n) {                         /* :423-427 */
      const double* phin = s->phi + (long)J * (n - 1);
      const double* un = s->u + (long)J * (n - 1);
      const double* Wp = s->W + (long)J * (n - 1);
      double acc = 0.0;
      for (int j = 0; j < J; ++j) {
        f[j] = phin[j] * (f[j] + Wp[j] * tmp);
        acc += un[j] * f[j];
      }
      tmp = sqrt(s->D[n]) * zk[n];
      yk[n] = tmp + acc;
    }
  }
  free(f);
  return REF_OK;
}

/* ---- dot: cholesky.h:444-590 ------------------------------------------------ */
int ref_dot(double jitter,
            int n_a_real, const double* a_real,
            int n_c_real, const double* c_real,
            int n_a_comp, const double* a_comp,
            int n_b_comp, const double* b_comp,
            int n_c_comp, const double* c_comp,
            int n_d_comp, const double* d_comp,
            int n_A, const double* A,
            int U_rows, int U_cols, const double* U,
            int V_rows, int V_cols, const double* V,
            int n_x, const double* x,
            int z_rows, int nrhs, const double* z, double* y)
{
  const int N = z_rows;
  if (n_x != z_rows) return REF_DIMENSION_MISMATCH;       /* :459-463 */
  if (n_a_real != n_c_real) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_b_comp) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_c_comp) return REF_DIMENSION_MISMATCH;
  if (n_a_comp != n_d_comp) return REF_DIMENSION_MISMATCH;
  const int has_general = (n_A != 0);                     /* :465-469 */
  if (has_general && n_A != N) return REF_DIMENSION_MISMATCH;
  if (has_general && U_cols != N) return REF_DIMENSION_MISMATCH;
  if (has_general && V_cols != N) return REF_DIMENSION_MISMATCH;
  if (U_rows != V_rows) return REF_DIMENSION_MISMATCH;

  const int J_general = U_rows, J_real = n_a_real, J_comp = n_a_comp;
  const int J = J_real + 2 * J_comp + J_general;

  if (J == 0) {                                           /* :477-481 */
    for (long i = 0; i < (long)N * nrhs; ++i) y[i] = jitter * z[i];
    return REF_OK;
  }

  double sum_ar = 0.0, sum_ac = 0.0;                      /* :483-485 */
  for (int j = 0; j < J_real; ++j) sum_ar += a_real[j];
  for (int j = 0; j < J_comp; ++j) sum_ac += a_comp[j];
  double* dg = (double*)malloc(sizeof(double) * (size_t)N);
  for (int n = 0; n < N; ++n) {
    dg[n] = (sum_ar + sum_ac) + jitter;
    if (has_general) dg[n] += A[n];
  }

  const long Nm1 = N - 1;
  double* phi = (double*)calloc((size_t)(J * Nm1 + 1), sizeof(double));
  double* u = (double*)calloc((size_t)(J * Nm1 + 1), sizeof(double));
  double* v = (double*)calloc((size_t)((long)J * N + 1), sizeof(double));

  for (int j = 0; j < J_real; ++j) v[j] = 1.0;            /* :492-499 */
  for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
    const double arg = d_comp[j] * x[0];
    v[k] = cos(arg);
    v[k + 1] = sin(arg);
  }
  for (int n = 0; n < N - 1; ++n) {                       /* :502-531 */
    const double dx = x[n + 1] - x[n];
    for (int j = 0; j < J_real; ++j) {
      v[j + (long)J * (n + 1)] = 1.0;
      u[j + (long)J * n] = a_real[j];
      phi[j + (long)J * n] = exp(-c_real[j] * dx);
    }
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const double a = a_comp[j], b = b_comp[j];
      const double arg = d_comp[j] * x[n + 1];
      const double cd = cos(arg), sd = sin(arg);
      v[k + (long)J * (n + 1)] = cd;
      v[k + 1 + (long)J * (n + 1)] = sd;
      u[k + (long)J * n] = a * cd + b * sd;
      u[k + 1 + (long)J * n] = a * sd - b * cd;
      phi[k + (long)J * n] = phi[k + 1 + (long)J * n] = exp(-c_comp[j] * dx);
    }
    for (int j = 0, k = J_real + 2 * J_comp; j < J_general; ++j, ++k) {
      /* note the reference's indexing: v at n, u at n+1 (:527-528); the last
       * column of v for general rows is left unset there and never read. */
      v[k + (long)J * n] = V[(long)j * N + n];
      u[k + (long)J * n] = U[(long)j * N + n + 1];
      phi[k + (long)J * n] = 1.0;
    }
  }

  double* f = (double*)malloc(sizeof(double) * (size_t)J);
  for (int kk = 0; kk < nrhs; ++kk) {                     /* :535-560 */
    const double* zk = z + (long)kk * N;
    double* yk = y + (long)kk * N;
    yk[N - 1] = dg[N - 1] * zk[N - 1];
    for (int j = 0; j < J; ++j) f[j] = 0.0;
    for (int n = N - 2; n >= 0; --n) {
      const double z0 = zk[n + 1];
      double y0 = dg[n] * zk[n];
      for (int j = 0; j < J; ++j) {
        const double value = phi[j + (long)J * n] * (f[j] + u[j + (long)J * n] * z0);
        f[j] = value;
        y0 += v[j + (long)J * n] * value;
      }
      yk[n] = y0;
    }
    for (int j = 0; j < J; ++j) f[j] = 0.0;
    for (int n = 1; n < N; ++n) {
      const double z0 = zk[n - 1];
      double y0 = yk[n];
      for (int j = 0; j < J; ++j) {
        const double value =
            phi[j + (long)J * (n - 1)] * (f[j] + v[j + (long)J * (n - 1)] * z0);
        f[j] = value;
        y0 += u[j + (long)J * (n - 1)] * value;
      }
      yk[n] = y0;
    }
  }
  free(f); free(phi); free(u); free(v); free(dg);
  return REF_OK;
}

/* ---- predict: cholesky.h:599-698 -------------------------------------------- */
int ref_predict(const ref_solver* s, int y_rows, const double* y,
                int M, const double* xs, double* pred)
{
  if (y_rows != s->N) return REF_DIMENSION_MISMATCH;      /* :600 */
  if (!s->computed) return REF_NOT_COMPUTED;              /* :601 */
  const int N = s->N, J = s->J, J_real = s->J_real, J_comp = s->J_comp;
  const double* t_ = s->t;

  double* alpha = (double*)malloc(sizeof(double) * (size_t)N);
  int st = ref_solve(s, N, 1, y, alpha);                  /* :608 */
  if (st != REF_OK) { free(alpha); return st; }
  for (int m = 0; m < M; ++m) pred[m] = 0.0;
  double* Q = (double*)calloc((size_t)(J > 0 ? J : 1), sizeof(double));

  int m = 0;                                              /* :615-653 */
  while (m < M && xs[m] <= t_[0]) ++m;
  for (int n = 0; n < N; ++n) {
    const double alphan = alpha[n];
    const double tref = (n < N - 1) ? t_[n + 1] : t_[N - 1];
    const double tn = t_[n];
    double dt = tref - tn;
    for (int j = 0; j < J_real; ++j) {
      Q[j] += alphan;
      Q[j] *= exp(-s->c_real[j] * dt);
    }
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const double tmp = exp(-s->c_comp[j] * dt);
      Q[k] += alphan * cos(s->d_comp[j] * tn);
      Q[k] *= tmp;
      Q[k + 1] += alphan * sin(s->d_comp[j] * tn);
      Q[k + 1] *= tmp;
    }
    while (m < M && (n == N - 1 || xs[m] <= tref)) {
      const double xm = xs[m];
      dt = xm - tref;
      double pm = 0.0;
      for (int j = 0; j < J_real; ++j)
        pm += s->a_real[j] * exp(-s->c_real[j] * dt) * Q[j];
      for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
        const double cd = cos(s->d_comp[j] * xm), sd = sin(s->d_comp[j] * xm);
        const double tmp = exp(-s->c_comp[j] * dt);
        pm += (s->a_comp[j] * cd + s->b_comp[j] * sd) * tmp * Q[k];
        pm += (s->a_comp[j] * sd - s->b_comp[j] * cd) * tmp * Q[k + 1];
      }
      pred[m] = pm;
      ++m;
    }
  }

  m = M - 1;                                              /* :656-695 */
  while (m >= 0 && xs[m] > t_[N - 1]) --m;
  for (int j = 0; j < J; ++j) Q[j] = 0.0;
  for (int n = N - 1; n >= 0; --n) {
    const double alphan = alpha[n];
    const double tref = (n > 0) ? t_[n - 1] : t_[0];
    const double tn = t_[n];
    double dt = tn - tref;
    for (int j = 0; j < J_real; ++j) {
      Q[j] += alphan * s->a_real[j];
      Q[j] *= exp(-s->c_real[j] * dt);
    }
    for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
      const double cd = cos(s->d_comp[j] * tn), sd = sin(s->d_comp[j] * tn);
      const double tmp = exp(-s->c_comp[j] * dt);
      Q[k] += alphan * (s->a_comp[j] * cd + s->b_comp[j] * sd);
      Q[k] *= tmp;
      Q[k + 1] += alphan * (s->a_comp[j] * sd - s->b_comp[j] * cd);
      Q[k + 1] *= tmp;
    }
    while (m >= 0 && (n == 0 || xs[m] > tref)) {
      const double xm = xs[m];
      dt = tref - xm;
      double pm = 0.0;
      for (int j = 0; j < J_real; ++j) pm += exp(-s->c_real[j] * dt) * Q[j];
      for (int j = 0, k = J_real; j < J_comp; ++j, k += 2) {
        const double tmp = exp(-s->c_comp[j] * dt);
        pm += cos(s->d_comp[j] * xm) * tmp * Q[k];
        pm += sin(s->d_comp[j] * xm) * tmp * Q[k + 1];
      }
      pred[m] += pm;
      --m;
    }
  }
  free(Q); free(alpha);
  return REF_OK;
}

/* ---- log-likelihood wrapper: celerite/celerite.py:180-219 ------------------- */
int ref_log_likelihood(double jitter,
                       int J_real, const double* a_real, const double* c_real,
                       int J_comp, const double* a_comp, const double* b_comp,
                       const double* c_comp, const double* d_comp,
                       int N, const double* t, const double* diag, const double* y,
                       double* loglike, double* logdet, double* quad)
{
  ref_solver* s = ref_create();
  int st = ref_compute(s, jitter, J_real, a_real, J_real, c_real, J_comp, a_comp,
                       J_comp, b_comp, J_comp, c_comp, J_comp, d_comp,
                       0, NULL, 0, 0, NULL, 0, 0, NULL, N, t, N, diag);
  if (st != REF_OK) {
    *loglike = -INFINITY; *logdet = NAN; *quad = NAN;     /* quiet=True: :205-208 */
    ref_destroy(s);
    return st;
  }
  double ld = s->log_det, q = NAN, ll;
  if (!isfinite(ld)) {                                    /* :212-213 */
    ll = -INFINITY;
  } else {
    ref_dot_solve(s, N, y, &q);
    ll = -0.5 * (q + ld + N * log(2.0 * M_PI));           /* :214-216 */
    if (!isfinite(ll)) ll = -INFINITY;                    /* :217-218 */
  }
  *loglike = ll; *logdet = ld; *quad = q;
  ref_destroy(s);
  return REF_OK;
}

/* ---- batched baseline -------------------------------------------------------- */
typedef struct {
  int B, N, J_real, J_comp, tid, nthreads;
  const double *jitter, *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp;
  const double *t, *diag, *y;
  long t_stride, diag_stride, y_stride;
  double *loglike, *logdet, *quad;
  int* status;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* jb = (batch_job*)arg;
  for (int p = jb->tid; p < jb->B; p += jb->nthreads) {
    jb->status[p] = ref_log_likelihood(
        jb->jitter[p], jb->J_real, jb->a_real + (long)p * jb->J_real,
        jb->c_real + (long)p * jb->J_real, jb->J_comp,
        jb->a_comp + (long)p * jb->J_comp, jb->b_comp + (long)p * jb->J_comp,
        jb->c_comp + (long)p * jb->J_comp, jb->d_comp + (long)p * jb->J_comp,
        jb->N, jb->t + p * jb->t_stride, jb->diag + p * jb->diag_stride,
        jb->y + p * jb->y_stride, jb->loglike + p, jb->logdet + p, jb->quad + p);
  }
  return NULL;
}

int ref_batch_log_likelihood(int B, int N, int J_real, int J_comp,
                             const double* jitter,
                             const double* a_real, const double* c_real,
                             const double* a_comp, const double* b_comp,
                             const double* c_comp, const double* d_comp,
                             const double* t, long t_stride,
                             const double* diag, long diag_stride,
                             const double* y, long y_stride,
                             double* loglike, double* logdet, double* quad,
                             int* status, int nthreads)
{
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  batch_job jobs[256];
  pthread_t th[256];
  for (int i = 0; i < nthreads; ++i) {
    batch_job jb = {B, N, J_real, J_comp, i, nthreads, jitter, a_real, c_real,
                    a_comp, b_comp, c_comp, d_comp, t, diag, y, t_stride,
                    diag_stride, y_stride, loglike, logdet, quad, status};
    jobs[i] = jb;
  }
  if (nthreads == 1) {
    batch_worker(&jobs[0]);
    return REF_OK;
  }
  for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, batch_worker, &jobs[i]);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  return REF_OK;
}
