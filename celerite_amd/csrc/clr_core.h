// celerite_amd/csrc/clr_core.h
//
// Per-lane arithmetic of the batched semiseparable-Cholesky log-likelihood for
// small widths (J = J_real + 2 J_comp <= 8): everything one GPU lane does for
// one (problem, chunk-of-the-time-axis) pair, with all state in registers.
//
// The reference (cpp/include/celerite/solver/cholesky.h:126-179, :348-357)
// walks n = 1..N-1 sequentially.  Here the time axis is cut into chunks and the
// recurrence is evaluated as a three-phase associative scan:
//
//   summarize  (parallel over chunks)  fold the chunk's steps, started from the
//              zero state, into one "transfer element" (A, b, C, eta, Jm) that
//              maps the state (P, f) at the chunk's first sample to the state at
//              the next chunk's first sample;
//   prefix     (sequential over a problem's chunks, parallel over problems)
//              apply the elements in order -> exact start state of every chunk;
//   correct    (parallel over chunks)  turn each chunk's zero-start sums into its
//              true log-det / quadratic contributions from its start state alone
//              (determinant lemma + Woodbury, chunk_update below) and certify
//              positive definiteness -- the fused log-likelihood needs no second
//              pass over the series;
//   replay     (parallel over chunks)  the reference recurrence itself, started
//              from that state: only for problems the certificate could not
//              settle, and to write the factor phi, u, W, D to HBM when asked.
//
// State convention.  "State at sample n" = (P_n, f_n) BEFORE sample n is used:
//   P_n = the reference's S after its update at step n (cholesky.h:154-160),
//   f_n = the reference's f after its update at step n (cholesky.h:350-352).
// One step (use sample n, then move to n+1):
//   q = P u_n ; D_n = a_n - u_n.q ; z = v_n - q ; W_n = z / D_n      (:162-178)
//   x_n = y_n - u_n.f                                                  (:349-355)
//   P_{n+1} = Phi (P + z z^T / D_n) Phi ; f_{n+1} = Phi (f + W_n x_n)  (:154-160, :351)
// with u_n = U~(t_n), v_n = V~(t_n), Phi = diag(exp(-c (t_{n+1} - t_n))).
//
// The chunk map is a linear-fractional (Riccati) map; in the form used for
// parallel Kalman filtering (Sarkka & Garcia-Fernandez, IEEE TAC 66 (2021)):
//   P' = C + A P (I + Jm P)^-1 A^T ,   f' = A (I + P Jm)^-1 (f + P eta) + b .
// A single step has A = Phi (I - v u^T / a), b = Phi v y / a, C = Phi v v^T Phi / a,
// eta = -u y / a, Jm = -u u^T / a.  Folding one step onto a running element
// (A, b, C, eta, Jm) needs no matrix inverse (Sherman-Morrison collapses it):
//   q = C u ; D = a - u.q ; z = v - q ; W = z / D ; x = y - u.b ; r = A^T u
//   eta -= r x / D ; Jm -= r r^T / D
//   C <- Phi (C + z z^T / D) Phi ; b <- Phi (b + W x) ; A <- Phi (A - W r^T)
// i.e. (C, b) is the zero-start trajectory and (A, eta, Jm) ride along.
//
// This header is plain C++14 usable from hipcc (device) and from g++ (the
// host-side algebra check in tests/hostcheck -- test infrastructure, not a
// product path: libcelerite_hip.so contains no CPU implementation).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define CLR_HD __host__ __device__ __forceinline__
#define CLR_UNROLL _Pragma("unroll")
// inside templates on the width J: full unrolling (register-resident arrays) up to width 8,
// rolled loops (arrays in scratch) for the padded widths 16 / 32 of the wide scan
#define CLR_UNROLL_J _Pragma("clang loop unroll_count(J <= 8 ? 64 : 1)")
#else
#define CLR_HD inline
#define CLR_UNROLL
#define CLR_UNROLL_J
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#endif

namespace clr {

// Packed upper triangle of a symmetric J x J matrix: (k <= j) -> k + j(j+1)/2.
CLR_HD constexpr int tri(int k, int j) { return k + j * (j + 1) / 2; }
CLR_HD constexpr int sym(int i, int j) { return i <= j ? tri(i, j) : tri(j, i); }
CLR_HD constexpr int nz(int n) { return n > 0 ? n : 1; }

template <int JR, int JC>
struct Widths {
  static constexpr int J = JR + 2 * JC;
  static constexpr int SZ = J * (J + 1) / 2;
  // doubles per transfer element / per chunk start state in the workspace
  static constexpr int ELEM = J * J + J + SZ + J + SZ;
  static constexpr int START = SZ + J;
};

// Kernel hyper-parameters of one problem (wave-uniform in the batch kernels).
template <int JR, int JC>
struct Problem {
  double ar[nz(JR)], cr[nz(JR)];
  double ac[nz(JC)], bc[nz(JC)], cc[nz(JC)], dc[nz(JC)];
  double sum_ar, sum_ac, jitter;

  CLR_HD void load(const double* a_real, const double* c_real, const double* a_comp,
                   const double* b_comp, const double* c_comp, const double* d_comp,
                   double jit) {
    sum_ar = 0.0;
    sum_ac = 0.0;
    CLR_UNROLL
    for (int j = 0; j < JR; ++j) { ar[j] = a_real[j]; cr[j] = c_real[j]; sum_ar += ar[j]; }
    CLR_UNROLL
    for (int j = 0; j < JC; ++j) {
      ac[j] = a_comp[j]; bc[j] = b_comp[j]; cc[j] = c_comp[j]; dc[j] = d_comp[j];
      sum_ac += ac[j];
    }
    jitter = jit;
  }
  // Diagonal of K at one sample, summed in the reference's order (cholesky.h:98).
  CLR_HD double diagonal(double diag_n) const { return ((diag_n + sum_ar) + sum_ac) + jitter; }
};

// How a lane reads the three input series for its chunk of L samples.  The chunk
// routines below are written against a small policy ("Src") so that the same
// arithmetic serves three data paths:
//   t(i), diag(i), y(i)   sample i of the lane's chunk (i <= L for t: the decay of
//                         the chunk's last step needs the next chunk's first time)
//   prologue(), step_begin(i), step_end(i)   hooks around every step; the loops run
//                         a WAVE-UNIFORM number of steps (L) so that a cooperative
//                         policy may load tiles and hit barriers inside them.
// DirectSeries: plain addressing, element (chunk c, local i) at base[c * cs + i * is]
//   row-major    ([problem][n], the layout of the public API)      is = 1,      cs = L
//   interleaved  ([problem][i][chunk], written by relayout_kernel)  is = nchunk, cs = 1
// StagedSeries (clr_batch_kernels.h, device only): the wave reads the row-major
//   arrays in coalesced 8-step tiles and transposes them through LDS.
struct DirectSeries {
  const double *tp, *dp, *yp;  // already offset to this lane's (problem, chunk) origin
  long is, cs;
  int L;
  long nleft;  // samples from this lane's origin to the end of the series (may be <= 0)
  CLR_HD long off(int i) const { return i < L ? (long)i * is : cs + (long)(i - L) * is; }
  CLR_HD double t(int i) const { return i < nleft ? tp[off(i)] : 0.0; }
  CLR_HD double diag(int i) const { return i < nleft ? dp[off(i)] : 0.0; }
  CLR_HD double y(int i) const { return i < nleft ? yp[off(i)] : 0.0; }
  CLR_HD void prologue() {}
  CLR_HD void step_begin(int) {}
  CLR_HD void step_end(int) {}
};

// ---------------------------------------------------------------------------
// sin and cos of the absolute phase d * t.  ocml's fp64 sincos spends 108 fp64
// instructions per call (double-double reduction and polynomial tails) to be
// correctly rounded RELATIVE to results near zero; the recurrence only needs
// ABSOLUTE accuracy ~1 ulp(1) because sin/cos enter U~, V~ linearly
// (cholesky.h:143-146).  This version: 3-term Cody-Waite reduction with FMA
// (exact products, so pi/2 is split into full 53-bit pieces), then the fdlibm
// minimax kernels on [-pi/4, pi/4].  Measured against long-double libm over
// |x| < 1e9: max abs error 1.8e-16 (tests/test_hostcheck.py).  There is no
// in-kernel fallback branch (an inlined ocml sincos under a branch costs ~60
// registers even when never taken): the HOST checks max|d| * max|t| -- O(B),
// t is sorted -- and launches the <FAST = false> instantiation (ocml sincos)
// when the product is >= 1e9 or not finite.
// ---------------------------------------------------------------------------
CLR_HD void sincos_fast(double x, double* s_out, double* c_out) {
  const double k = rint(x * 0.63661977236758134308);  // 2/pi
  double r = fma(-k, 1.57079632679489655800e+00, x);  // pi/2 = P1 + P2 + P3 (each fl())
  r = fma(-k, 6.12323399573676603587e-17, r);
  r = fma(-k, -1.49738490485916983043e-33, r);
  const double z = r * r;
  // fdlibm __kernel_sin / __kernel_cos coefficients
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sn = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)k & 3;
  const double s0 = (q & 1) ? cs : sn;
  const double c0 = (q & 1) ? sn : cs;
  *s_out = (q & 2) ? -s0 : s0;
  *c_out = ((q + 1) & 2) ? -c0 : c0;
}

// FAST is chosen on the host per launch: max|d_comp| * max|t| < CLR_FAST_TRIG_LIMIT.
#define CLR_FAST_TRIG_LIMIT 1.0e9
template <bool FAST>
CLR_HD void sincos_phase(double x, double* s_out, double* c_out) {
  if (FAST)
    sincos_fast(x, s_out, c_out);
  else
    sincos(x, s_out, c_out);
}

// 1 / D for the summarize kernels: v_rcp_f64 + two Newton steps (5 fp64 instructions, <= 1 ulp) instead of the
// IEEE division sequence (div_scale x2, rcp, 5 fma, div_fmas, div_fixup: 12).  D is a pivot of order 1e-6 .. 1e6, or
// the 1e300 of a padded step; D = 0 or a non-finite D gives NaN here instead of +-inf / 0 -- such a pivot is flagged
// by its caller (!(D > 0)) and the problem is settled by the reference recurrence, which keeps the IEEE division
// (replay_chunk writes W = z / D into the factor).  Host build (tests/hostcheck): the plain division.
CLR_HD double recip_fast(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
#else
  return 1.0 / d;
#endif
}

// Streaming store for the materialised factor (20 GB per launch at the headline shape, written once and not read
// back by the kernel): bypasses the caches on the device; a plain store on the host build.
CLR_HD void store_stream(double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CLR_NO_STREAM_STORES)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// sum of log(D_n) without a log per step: keep the product of the mantissas and
// the sum of the exponents (ocml's fp64 log is 76 fp64 instructions; this is 3).
// D = 0 -> product 0 -> log = -inf; D < 0 is flagged by the caller; NaN propagates
// -- the same non-finite outcomes as log(D_).sum() (cholesky.h:208, celerite.py:212).
struct LogProduct {
  double mant;
  int expo, since;
  CLR_HD void init() { mant = 1.0; expo = 0; since = 0; }
  CLR_HD void mul(double d) {
    int e;
    mant *= frexp(d, &e);
    expo += e;
    if (++since == 32) {  // 2^-32 at worst per renormalisation window: no underflow
      mant = frexp(mant, &e);
      expo += e;
      since = 0;
    }
  }
  // for callers with their own window of <= 32 factors (the lazy split kernel's 16-step blocks): no counter
  CLR_HD void mul_window(double d) {
    int e;
    mant *= frexp(d, &e);
    expo += e;
  }
  CLR_HD void renorm() {
    int e;
    mant = frexp(mant, &e);
    expo += e;
  }
  CLR_HD double log_value() const { return log(mant) + expo * 0.693147180559945309417232; }
};

// U~(t), V~(t): cholesky.h:129-147 (real rows: a, 1; complex pair: (a cd + b sd,
// a sd - b cd), (cd, sd) with the ABSOLUTE time in the phase, :137).
template <int JR, int JC, bool FAST>
CLR_HD void features_uv(const Problem<JR, JC>& p, double t, double* u, double* v) {
  CLR_UNROLL
  for (int j = 0; j < JR; ++j) { u[j] = p.ar[j]; v[j] = 1.0; }
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) {
    const int k = JR + 2 * j;
    double sd, cd;
    sincos_phase<FAST>(p.dc[j] * t, &sd, &cd);
    u[k] = p.ac[j] * cd + p.bc[j] * sd;
    u[k + 1] = p.ac[j] * sd - p.bc[j] * cd;
    v[k] = cd;
    v[k + 1] = sd;
  }
}

// Wave-level "all lanes agree" (host: the single lane's own condition).
#if defined(__HIP_DEVICE_COMPILE__)
#define CLR_WAVE_ALL(cond) (__all(cond) != 0)
#else
#define CLR_WAVE_ALL(cond) (cond)
#endif

// exp(x) for |x| < 2^-7 by a degree-6 Taylor polynomial: truncation x^7/7! < 3.5e-19
// relative, i.e. below half an ulp -- the result is the correctly rounded one up to
// the rounding of the 6 FMAs (same as the library's own polynomial).
CLR_HD double exp_small(double x) {
  double p = fma(x, 1.0 / 720.0, 1.0 / 120.0);
  p = fma(x, p, 1.0 / 24.0);
  p = fma(x, p, 1.0 / 6.0);
  p = fma(x, p, 0.5);
  p = fma(x, p, 1.0);
  return fma(x, p, 1.0);
}
// |x| < 2^-10: the quartic is enough (x^5/120 < 7.4e-18, a fifteenth of an ulp of 1)
CLR_HD double exp_tiny(double x) {
  double p = fma(x, 1.0 / 24.0, 1.0 / 6.0);
  p = fma(x, p, 0.5);
  p = fma(x, p, 1.0);
  return fma(x, p, 1.0);
}

// The distinct decay factors of one step: phid[0..JR) for the real terms, then one
// per complex PAIR (both rows of a pair share it, cholesky.h:140-142), for the move
// from t to t + dx (cholesky.h:130,140).  Densely sampled series (|c dx| < 2^-7 in
// every lane of the wave -- the large-N regime) take the 6-FMA polynomial instead of
// the library exp (19 fp64 instructions); the choice is wave-uniform, no divergence.
template <int JR, int JC>
CLR_HD void features_phi_distinct(const Problem<JR, JC>& p, double dx, double* phid) {
  double x[nz(JR + JC)];
  double amax = 0.0;
  CLR_UNROLL
  for (int j = 0; j < JR; ++j) { x[j] = -p.cr[j] * dx; amax = fmax(amax, fabs(x[j])); }
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) { x[JR + j] = -p.cc[j] * dx; amax = fmax(amax, fabs(x[JR + j])); }
  if (CLR_WAVE_ALL(amax < 0.0009765625)) {  // 2^-10: the N = 1e5 regime (c dx ~ 1e-4, gaps up to 10x the mean)
    CLR_UNROLL
    for (int j = 0; j < JR + JC; ++j) phid[j] = exp_tiny(x[j]);
  } else if (CLR_WAVE_ALL(amax < 0.0078125)) {
    CLR_UNROLL
    for (int j = 0; j < JR + JC; ++j) phid[j] = exp_small(x[j]);
  } else {
    CLR_UNROLL
    for (int j = 0; j < JR + JC; ++j) phid[j] = exp(x[j]);
  }
}

// row k of the width-J state -> index of its decay factor in phid
template <int JR>
CLR_HD constexpr int phi_index(int k) { return k < JR ? k : JR + (k - JR) / 2; }

// S <- Phi (S + z w^T) Phi on the packed upper triangle, with the (JR+JC)(JR+JC+1)/2
// distinct products phi_a phi_b formed once (cholesky.h:154-160 does 3 flops per
// entry; this is 1 fma + 1 mul per entry + 15 products at J = 8).
template <int JR, int JC>
CLR_HD void decay_rank1_update(const double* phid, const double* z, const double* w, double* S) {
  constexpr int J = JR + 2 * JC;
  constexpr int M = JR + JC;
  double pp[nz(M * (M + 1) / 2)];
  CLR_UNROLL
  for (int b = 0; b < M; ++b) {
    CLR_UNROLL
    for (int a = 0; a <= b; ++a) pp[tri(a, b)] = phid[a] * phid[b];
  }
  CLR_UNROLL
  for (int j = 0; j < J; ++j) {
    CLR_UNROLL
    for (int k = 0; k <= j; ++k)
      S[tri(k, j)] = pp[tri(phi_index<JR>(k), phi_index<JR>(j))] * fma(z[k], w[j], S[tri(k, j)]);
  }
}

// ---------------------------------------------------------------------------
// summarize: fold the L samples of one (full, non-final) chunk into a transfer
// element that ends at the next chunk's first sample.  elem layout:
//   A[J*J] row-major | b[J] | C[SZ] | eta[J] | Jm[SZ]
// ---------------------------------------------------------------------------
// DENSE (densely sampled series: max |d| max dx < 2^-5 over the whole plan, checked on the host -- lazy_eligible): the
// (cos, sin) pairs of the complex terms are ROTATED from sample to sample through the small angle d dx (12 fp64
// instructions per term; clr_split_kernels.h has the same step) and anchored by the full sincos of the absolute phase
// (cholesky.h:137) every 16 steps, instead of a range-reduced sincos per term and step (~45 instructions).
template <int JR, int JC, bool FAST, class Src, bool DENSE = false>
CLR_HD void summarize_chunk(const Problem<JR, JC>& p, Src& src, int L, int n0, int N, bool store,
                            double* elem_out, double* ld0_out, double* q0_out, int* flag0_out,
                            double* gamma_out = nullptr) {
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  // State: 152 doubles at J = 8, more than the 128 that 256 VGPRs hold; the compiler
  // parks the excess (in practice A) in AGPRs and pays one VALU slot per 32-bit
  // move.  Keeping part of it in LDS instead was measured SLOWER (exposed LDS
  // latency at one wave per SIMD): profiles/r01c_lds_state_ab.log.
  // A is held COLUMN-major (Acol[j * J + i] = A[i][j]): column j only ever needs
  // r_j = u . A[:, j], so each column is read, used and rewritten in one visit.
  double Acol[J * J], b[J], C[SZ], eta[J], Jm[SZ];
  CLR_UNROLL
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL
    for (int j = 0; j < J; ++j) Acol[j * J + i] = (i == j) ? 1.0 : 0.0;
    b[i] = 0.0;
    eta[i] = 0.0;
  }
  CLR_UNROLL
  for (int i = 0; i < SZ; ++i) { C[i] = 0.0; Jm[i] = 0.0; }

  // Every lane runs L steps (the loop count must be wave-uniform for the staged
  // source).  Steps at or beyond the end of the series (the short last chunk, lanes
  // past the last chunk) run on padding: they must not touch the accumulate-only
  // outputs (Jm, eta, the zero-start sums); what they do to A, C, b is irrelevant
  // because the last chunk's element is never applied.
  const int len = L;
  double q0 = 0.0;
  double gamma = 0.0;  // max a_n / D_n of the zero-start pivots: the cancellation in D = a - u.Pu
  LogProduct lp0;
  lp0.init();
  int flag0 = 0;
  src.prologue();
  double tn = src.t(0);
  double t_next = src.t(1), diag_n = src.diag(0), y_n = src.y(0);
  double cdv[nz(JC)], sdv[nz(JC)];  // (DENSE) cos / sin of d t at the current sample
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) { cdv[j] = 1.0; sdv[j] = 0.0; }
  for (int i = 0; i < len; ++i) {
    src.step_begin(i);
    // register prefetch of the next sample
    const double t_cur_next = t_next, diag_cur = diag_n, y_cur = y_n;
    if (i + 1 < len) {
      t_next = src.t(i + 2);
      diag_n = src.diag(i + 1);
      y_n = src.y(i + 1);
    }

    double u[J], v[J], phid[nz(JR + JC)];
    if (DENSE) {
      if ((i & 15) == 0) {  // (wave-uniform) anchor: the absolute phase
        CLR_UNROLL
        for (int j = 0; j < JC; ++j) sincos_phase<FAST>(p.dc[j] * tn, &sdv[j], &cdv[j]);
      }
      CLR_UNROLL
      for (int j = 0; j < JR; ++j) { u[j] = p.ar[j]; v[j] = 1.0; }
      CLR_UNROLL
      for (int j = 0; j < JC; ++j) {
        const int k = JR + 2 * j;
        u[k] = p.ac[j] * cdv[j] + p.bc[j] * sdv[j];
        u[k + 1] = p.ac[j] * sdv[j] - p.bc[j] * cdv[j];
        v[k] = cdv[j];
        v[k + 1] = sdv[j];
        // ... and on to the next sample
        const double dl = p.dc[j] * (t_cur_next - tn), d2 = dl * dl;
        const double sn = dl * fma(d2, fma(d2, fma(d2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
        const double cs = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
        const double c0 = cdv[j], s0 = sdv[j];
        cdv[j] = fma(c0, cs, -s0 * sn);
        sdv[j] = fma(s0, cs, c0 * sn);
      }
    } else {
      features_uv<JR, JC, FAST>(p, tn, u, v);
    }
    features_phi_distinct<JR, JC>(p, t_cur_next - tn, phid);

    double q[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc += C[sym(k, j)] * u[k];
      q[j] = acc;
    }
    double s = 0.0, ub = 0.0;
    CLR_UNROLL
    for (int j = 0; j < J; ++j) { s += u[j] * q[j]; ub += u[j] * b[j]; }
    const double D = p.diagonal(diag_cur) - s;
    const double invD = recip_fast(D);
    const double x = y_cur - ub;
    const bool valid = n0 + i < N;
    // zero-start sums of this chunk (corrected for the true start state by
    // chunk_update); a zero-start pivot <= 0 means the chunk's own block of K is
    // not positive definite: the problem is sent to the exact replay
    if (valid) {
      if (n0 + i >= 1 && !(D > 0.0)) flag0 = 1;
      lp0.mul(D);
      q0 += x * x * invD;
      gamma = fmax(gamma, fabs(p.diagonal(diag_cur) * invD));
    }
    const double xs = x * invD;

    double z[J], W[J], pw[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      z[j] = v[j] - q[j];
      W[j] = z[j] * invD;
      pw[j] = phid[phi_index<JR>(j)] * W[j];
    }
    // one visit per column of A: r_j = u . A[:, j] ; A[:, j] <- Phi (A[:, j] - W r_j)
    double r[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double racc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) racc += Acol[j * J + k] * u[k];
      r[j] = racc;
      CLR_UNROLL
      for (int k = 0; k < J; ++k)
        Acol[j * J + k] = phid[phi_index<JR>(k)] * Acol[j * J + k] - pw[k] * racc;
    }
    if (valid) {  // (a select, not a multiply by 0: padding steps may carry inf / NaN)
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        const double rsj = r[j] * invD;
        eta[j] -= r[j] * xs;
        CLR_UNROLL
        for (int k = 0; k <= j; ++k) Jm[tri(k, j)] -= r[k] * rsj;
      }
    }
    CLR_UNROLL
    for (int j = 0; j < J; ++j) b[j] = phid[phi_index<JR>(j)] * (b[j] + W[j] * x);
    decay_rank1_update<JR, JC>(phid, z, W, C);
    tn = t_cur_next;
    src.step_end(i);
  }
  if (!store) return;
  *ld0_out = lp0.log_value();
  *q0_out = q0;
  *flag0_out = flag0;
  if (gamma_out) *gamma_out = gamma;

  double* o = elem_out;  // A is written row-major
  CLR_UNROLL
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL
    for (int j = 0; j < J; ++j) o[i * J + j] = Acol[j * J + i];
  }
  o += J * J;
  CLR_UNROLL
  for (int i = 0; i < J; ++i) o[i] = b[i];
  o += J;
  CLR_UNROLL
  for (int i = 0; i < SZ; ++i) o[i] = C[i];
  o += SZ;
  CLR_UNROLL
  for (int i = 0; i < J; ++i) o[i] = eta[i];
  o += J;
  CLR_UNROLL
  for (int i = 0; i < SZ; ++i) o[i] = Jm[i];
}

// ---------------------------------------------------------------------------
// prefix: one chunk of the sequential phase.  Given the state (P, f) at the
// chunk's first sample and the chunk's element (A, b, C, eta, Jm):
//
//  (1) the chunk's TRUE contributions to log det K and b^T K^-1 b from its
//      zero-start sums, WITHOUT replaying it.  With N = -Jm = O^T D0^-1 O >= 0 and
//      eta = -O^T D0^-1 x0 (O: the chunk's whitened observation map, rows r_n^T):
//        sum log D   = sum log D0 + log det(I + Jm P)            (determinant lemma)
//        sum x^2 / D = sum x0^2 / D0 + 2 eta.f - f.Jm f + w.G w  (Woodbury)
//        w = Jm f - eta ,  G = P (I + Jm P)^-1
//      (check: one step gives log(a - u.Pu) and (y - u.f)^2 / (a - u.Pu));
//  (2) the state at the next chunk's first sample:
//        P' = C + A G A^T ,  f' = A (I + P Jm)^-1 (f + P eta) + b .
//
// VALUES come from one Gauss-Jordan elimination with partial pivoting on
// [M^T | P | h], M^T = I + P Jm, h = f + P eta (rows are exchanged with selects so
// every index stays a compile-time constant): G = M^-T P, g = M^-T h, and det M from
// the pivots.  N (entries ~1e5) and P (graded, with tiny directions) multiply to
// O(1) eigenvalues; the LU of M keeps 1e-13 accuracy there, whereas factorising N or
// P first (a symmetric route through I - F^T P F) was measured to lose 6+ digits.
//
// CERTIFICATE.  The reference throws when a true pivot D_n < 0 (cholesky.h:176).
// The chunk has only positive true pivots iff its block conditioned on the past is
// positive definite iff every eigenvalue 1 - lambda_i(N P) of M is positive.  det M
// > 0 alone would miss an even number of negative ones, so positivity is certified
// by a sufficient test that needs no accuracy: with F F^T = N + delta I (un-pivoted
// Cholesky is stable for the regularised matrix, and over-estimating N only makes
// the test stricter), I - F^T P F must have a Cholesky factorisation with pivots
// > 1e-5.  The smallest pivot mu also estimates the conditioning of M: the chunk is
// marked SUSPICIOUS -- and the caller runs the exact replay for that problem -- when
// the certificate fails, when det M <= 0 or when anything is non-finite.  The rounding-error
// estimate of the corrections, J eps / mu (times |w.G w| for the quadratic form: err_out), goes to
// the caller, which sums it over the problem's chunks and holds it against 3e-12 of the
// problem's log det / quadratic form (decide_kernel; round 4 -- rounds 1..3 held it against the
// CHUNK's own contribution, which with hundreds of short chunks per problem is near zero for one
// of them more often than not: profiles/r04w_chunk_error_budget.txt); without err_out the old
// per-chunk test applies.
// Single-lane form (host check, single-lane prefix kernel); prefix_coop_kernel
// distributes the same algebra over 16 lanes.
// ---------------------------------------------------------------------------
template <int J>
CLR_HD double pd_certificate(const double* P /*[SZ]*/, const double* Jm /*[SZ]*/) {
  // returns the smallest Cholesky pivot of I - F^T P F (<= 0 when it breaks down)
  double F[J][J], S[J][J];
  double nmax = 0.0;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) nmax = fmax(nmax, -Jm[tri(i, i)]);
  const double delta = 4e-13 * nmax;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) S[i][j] = -Jm[sym(i, j)] + ((i == j) ? delta : 0.0);
  }
  CLR_UNROLL_J
  for (int k = 0; k < J; ++k) {
    const double rs = 1.0 / sqrt(S[k][k]);
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) F[i][k] = (i >= k) ? S[i][k] * rs : 0.0;
    CLR_UNROLL_J
    for (int i = k + 1; i < J; ++i) {
      CLR_UNROLL_J
      for (int j = k + 1; j <= i; ++j) {
        S[i][j] -= F[i][k] * F[j][k];
        S[j][i] = S[i][j];
      }
    }
  }
  double PF[J][J], E[J][J];
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL_J
    for (int k = 0; k < J; ++k) {
      double acc = 0.0;
      CLR_UNROLL_J
      for (int m = k; m < J; ++m) acc += P[sym(i, m)] * F[m][k];
      PF[i][k] = acc;
    }
  }
  CLR_UNROLL_J
  for (int j = 0; j < J; ++j) {
    CLR_UNROLL_J
    for (int k = 0; k <= j; ++k) {
      double acc = (j == k) ? 1.0 : 0.0;
      CLR_UNROLL_J
      for (int i = k; i < J; ++i) acc -= F[i][k] * PF[i][j];
      E[j][k] = acc;
    }
  }
  double mu = 1.0;
  bool broke = false;
  CLR_UNROLL_J
  for (int k = 0; k < J; ++k) {
    const double d = E[k][k];
    if (!(d > 0.0)) broke = true;
    mu = (d < mu) ? d : mu;
    const double rs = 1.0 / sqrt(d);
    CLR_UNROLL_J
    for (int i = k; i < J; ++i) E[i][k] *= rs;
    CLR_UNROLL_J
    for (int i = k + 1; i < J; ++i) {
      CLR_UNROLL_J
      for (int j = k + 1; j <= i; ++j) E[i][j] -= E[i][k] * E[j][k];
    }
  }
  return broke ? -1.0 : mu;
}

template <int J>
CLR_HD void chunk_update(const double* elem, double* P /*[SZ]*/, double* f /*[J]*/, bool correct,
                         bool advance, double ld0, double q0, double* dld, double* dq,
                         int* suspicious, double* mu_out = nullptr, bool check_quad = true,
                         double* eg_out = nullptr, double* err_out = nullptr) {
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int NC = 2 * J + 1;  // [ M^T | P | h ]
  const double* A = elem;
  const double* b = elem + J * J;
  const double* C = b + J;
  const double* eta = C + SZ;
  const double* Jm = eta + J;

  double T[J][NC];
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double h = f[i];
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = (i == j) ? 1.0 : 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += P[sym(i, k)] * Jm[sym(k, j)];
      T[i][j] = acc;
      T[i][J + j] = P[sym(i, j)];
      h += P[sym(i, j)] * eta[j];
    }
    T[i][2 * J] = h;
  }

  double det = 1.0;
  CLR_UNROLL_J
  for (int col = 0; col < J; ++col) {
    int piv = col;
    double best = fabs(T[col][col]);
    CLR_UNROLL_J
    for (int i = col + 1; i < J; ++i) {
      const double cand = fabs(T[i][col]);
      const bool take = cand > best;
      best = take ? cand : best;
      piv = take ? i : piv;
    }
    CLR_UNROLL_J
    for (int c = col; c < NC; ++c) {
      double top = T[col][c];
      const double old_top = top;
      CLR_UNROLL_J
      for (int i = col + 1; i < J; ++i) {
        const bool hit = (i == piv);
        top = hit ? T[i][c] : top;
        T[i][c] = hit ? old_top : T[i][c];
      }
      T[col][c] = top;
    }
    det *= (piv != col) ? -T[col][col] : T[col][col];
    const double inv = 1.0 / T[col][col];
    CLR_UNROLL_J
    for (int c = col + 1; c < NC; ++c) T[col][c] *= inv;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      if (i == col) continue;
      const double m = T[i][col];
      CLR_UNROLL_J
      for (int c = col + 1; c < NC; ++c) T[i][c] -= m * T[col][c];
    }
  }
  // now T[i][J + j] = G[i][j] (symmetrised below), T[i][2J] = g[i]

  if (eg_out) {
    // REALIZED accuracy of G = (I + P Jm)^-1 P, measured instead of bounded by 1 / mu, on two probe vectors
    // (all ones; alternating signs): residual r = P z - G z - P Jm (G z), first-order forward error
    // dg = (I + P Jm)^-1 r = (I - G Jm) r; returned as max |dg| / max |G z| over the probes (the floor of
    // this estimate is the rounding of r itself, ~ J eps).  Vectors only: 12 J^2 flops, no J x J temporaries.
    double worst = 0.0;
    CLR_UNROLL
    for (int probe = 0; probe < 2; ++probe) {
      double gz[J], t1[J], r[J];
      double gmax = 0.0, emax = 0.0;
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double acc = 0.0;
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc += (probe && (k & 1)) ? -T[i][J + k] : T[i][J + k];
        gz[i] = acc;
        gmax = fmax(gmax, fabs(acc));
      }
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double acc = 0.0;
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc += Jm[sym(i, k)] * gz[k];
        t1[i] = acc;
      }
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double acc = -gz[i];
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc += P[sym(i, k)] * (((probe && (k & 1)) ? -1.0 : 1.0) - t1[k]);
        r[i] = acc;
      }
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double acc = 0.0;
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc += Jm[sym(i, k)] * r[k];
        t1[i] = acc;
      }
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double acc = r[i];
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc -= T[i][J + k] * t1[k];
        emax = (acc != acc) ? INFINITY : fmax(emax, fabs(acc));
      }
      const double e = (gmax > 0.0) ? emax / gmax : (emax == 0.0 ? 0.0 : INFINITY);
      worst = (e > worst || e != e) ? e : worst;
    }
    *eg_out = worst;
  }

  if (correct) {
    int bad = 0;
    const double mu = pd_certificate<J>(P, Jm);
    if (mu_out) *mu_out = mu;
    if (!(mu > 1e-5)) bad = 1;
    if (!(det > 0.0)) bad = 1;
    double w[J];
    double ef = 0.0, fJf = 0.0;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      double acc = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += Jm[sym(i, k)] * f[k];
      w[i] = acc - eta[i];
      ef += eta[i] * f[i];
      fJf += f[i] * acc;
    }
    double wGw = 0.0;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      double acc = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += 0.5 * (T[i][J + k] + T[k][J + i]) * w[k];
      wGw += w[i] * acc;
    }
    const double q = 2.0 * ef - fJf + wGw;
    const double ld = log(det);
    const double err = J * 2.2e-16 / mu;  // rounding-error estimate of the corrections
    if (err_out) {
      // the caller holds the estimates against the PROBLEM's log det and quadratic form (decide_kernel): a chunk's own
      // contribution can be anywhere near zero -- with hundreds of short chunks per problem one of them usually is
      *err_out = check_quad ? err * fabs(wGw) : 0.0;  // (the log det's share is J eps / mu: the caller has mu)
    } else {
      if (!(err <= 3e-12 * fabs(ld0 + ld))) bad = 1;
      if (check_quad && !(err * fabs(wGw) <= 3e-12 * fabs(q0 + q))) bad = 1;
    }
    if ((check_quad && !isfinite(q)) || !isfinite(ld)) bad = 1;
    *dld = ld;
    *dq = q;
    *suspicious = bad;
  }

  if (advance) {
    double AG[J][J];
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      CLR_UNROLL_J
      for (int j = 0; j < J; ++j) {
        double acc = 0.0;
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc += A[i * J + k] * (0.5 * (T[k][J + j] + T[j][J + k]));
        AG[i][j] = acc;
      }
    }
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL_J
      for (int k = 0; k <= j; ++k) {
        double acc = C[tri(k, j)];
        CLR_UNROLL_J
        for (int i = 0; i < J; ++i) acc += AG[k][i] * A[j * J + i];
        P[tri(k, j)] = acc;
      }
    }
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      double acc = b[i];
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += A[i * J + k] * T[k][2 * J];
      f[i] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// compose: the element of "e1, then e2" (the associative operator of the scan; Sarkka &
// Garcia-Fernandez 2021, Lemma 8, in this file's sign conventions).  With
// Mi = (I + C1 Jm2)^-1 and w = eta2 - Jm2 b1:
//   C12 = C2 + A2 (Mi C1) A2^T          b12   = b2 + A2 Mi (b1 + C1 eta2)
//   A12 = A2 (Mi A1)                    eta12 = eta1 + (Mi A1)^T w
//   Jm12 = Jm1 + A1^T Jm2 (Mi A1)
// (C12, b12) is nothing but e2 ADVANCING the state (C1, b1) -- (C, b) of an element is its
// zero-start trajectory -- so a composition is an advance (chunk_update) whose Gauss-Jordan
// tableau [I + C1 Jm2 | C1 | h] carries J more right-hand sides, the columns of A1:
// X1 = Mi A1 feeds the three rider updates.  (Mi^T Jm2 = Jm2 Mi: push-through identity.)
// Single-lane form: the host check, and the on-device cross-check of the cooperative
// group_compose_kernel (clr_prefix_kernels.h), which a multi-level prefix is built from.
// `out` may alias e1 or e2.
// ---------------------------------------------------------------------------
template <int J>
CLR_HD void compose_elements(const double* e1, const double* e2, double* out) {
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int NC = 3 * J + 1;  // [ I + C1 Jm2 | C1 | A1 | h ]
  const double *A1 = e1, *b1 = e1 + J * J, *C1 = b1 + J, *eta1 = C1 + SZ, *Jm1 = eta1 + J;
  const double *A2 = e2, *b2 = e2 + J * J, *C2 = b2 + J, *eta2 = C2 + SZ, *Jm2 = eta2 + J;

  double T[J][NC];
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double h = b1[i];
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = (i == j) ? 1.0 : 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += C1[sym(i, k)] * Jm2[sym(k, j)];
      T[i][j] = acc;
      T[i][J + j] = C1[sym(i, j)];
      T[i][2 * J + j] = A1[i * J + j];
      h += C1[sym(i, j)] * eta2[j];
    }
    T[i][3 * J] = h;
  }
  CLR_UNROLL_J
  for (int col = 0; col < J; ++col) {
    int piv = col;
    double best = fabs(T[col][col]);
    CLR_UNROLL_J
    for (int i = col + 1; i < J; ++i) {
      const double cand = fabs(T[i][col]);
      const bool take = cand > best;
      best = take ? cand : best;
      piv = take ? i : piv;
    }
    CLR_UNROLL_J
    for (int c = col; c < NC; ++c) {
      double top = T[col][c];
      const double old_top = top;
      CLR_UNROLL_J
      for (int i = col + 1; i < J; ++i) {
        const bool hit = (i == piv);
        top = hit ? T[i][c] : top;
        T[i][c] = hit ? old_top : T[i][c];
      }
      T[col][c] = top;
    }
    const double inv = 1.0 / T[col][col];
    CLR_UNROLL_J
    for (int c = col + 1; c < NC; ++c) T[col][c] *= inv;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      if (i == col) continue;
      const double m = T[i][col];
      CLR_UNROLL_J
      for (int c = col + 1; c < NC; ++c) T[i][c] -= m * T[col][c];
    }
  }
  // T[i][J + j] = X2 = Mi C1 (symmetric up to rounding), T[i][2J + j] = X1 = Mi A1, T[i][3J] = Mi h
  double w[J], A12[J * J], b12[J], C12[SZ], eta12[J], Jm12[SZ];
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double acc = eta2[i];
    CLR_UNROLL_J
    for (int k = 0; k < J; ++k) acc -= Jm2[sym(i, k)] * b1[k];
    w[i] = acc;
  }
  double Y[J][J];  // A2 X2 (then A2 X2 A2^T), reused for Jm2 X1
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double bacc = b2[i];
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = 0.0, aacc = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) {
        acc += A2[i * J + k] * (0.5 * (T[k][J + j] + T[j][J + k]));
        aacc += A2[i * J + k] * T[k][2 * J + j];
      }
      Y[i][j] = acc;
      A12[i * J + j] = aacc;
      bacc += A2[i * J + j] * T[j][3 * J];
    }
    b12[i] = bacc;
  }
  CLR_UNROLL_J
  for (int j = 0; j < J; ++j) {
    CLR_UNROLL_J
    for (int k = 0; k <= j; ++k) {
      double acc = C2[tri(k, j)];
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) acc += Y[k][i] * A2[j * J + i];
      C12[tri(k, j)] = acc;
    }
  }
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {  // Y <- Jm2 X1
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc += Jm2[sym(i, k)] * T[k][2 * J + j];
      Y[i][j] = acc;
    }
  }
  CLR_UNROLL_J
  for (int j = 0; j < J; ++j) {
    double eacc = eta1[j];
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) eacc += T[i][2 * J + j] * w[i];
    eta12[j] = eacc;
    CLR_UNROLL_J
    for (int k = 0; k <= j; ++k) {  // symmetrised: (A1^T Y + Y^T A1) / 2
      double acc = 0.0;
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) acc += A1[i * J + k] * Y[i][j] + A1[i * J + j] * Y[i][k];
      Jm12[tri(k, j)] = Jm1[tri(k, j)] + 0.5 * acc;
    }
  }
  double* o = out;
  CLR_UNROLL_J
  for (int i = 0; i < J * J; ++i) o[i] = A12[i];
  o += J * J;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) o[i] = b12[i];
  o += J;
  CLR_UNROLL_J
  for (int i = 0; i < SZ; ++i) o[i] = C12[i];
  o += SZ;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) o[i] = eta12[i];
  o += J;
  CLR_UNROLL_J
  for (int i = 0; i < SZ; ++i) o[i] = Jm12[i];
}

// ---------------------------------------------------------------------------
// Schedule of a multi-level prefix over n chunk elements: level 0 = the chunks, level l + 1 =
// compositions of groups of g[l] consecutive level-l elements; the top level is walked
// sequentially, then the start states fan out level by level (clr_prefix_kernels.h).
// Chosen on the host from a time model fitted on MI355X (profiles/r03a_prefix_ab.txt, r03p_prefix_ab.txt):
//   * one advance of a lone wave: t = 0.49 + 0.04 J^2 us (1.13 us at width 4, 3.04 at width 8);
//     a composition 1.3 t (same Gauss-Jordan on two DPP rows, one more J^3 / 16 product);
//   * a phase of W waves runs in max(1, W / (1024 SIMDs x k)) rounds, k = min(waves the kernel's registers
//     allow per SIMD, overlap): up to width 5 three resident waves overlap almost for free (a lone wave
//     issues one instruction per ~5 cycles); at widths 6..8 the chains are dense fp64 and a second wave on
//     the SIMD buys only ~1.3x (fan-out of 2048 waves at width 8: 6.6 us per step against 3.0 alone);
//   * every level adds two dependent launches (~6 us each).
// With the running composition in LDS (190 registers, two waves per SIMD at width 8) one level of groups of
// 8 takes 0.146 ms at B = 1024 x 64 chunks against 0.179 ms for the walk; 256 x 125 chunks at width 4 runs
// 2.5x faster than the walk.
// ---------------------------------------------------------------------------
struct PrefixPlan {
  int levels;     // number of composition levels (0: plain sequential walk)
  int g[3];       // group sizes, bottom up
  int n[4];       // element counts per level: n[0] = nchunk, n[l + 1] = ceil(n[l] / g[l])
  double time_us; // modelled duration of the prefix phase
};
inline double prefix_plan_time_us(const PrefixPlan& p, int B, int J) {
  static const int occ_compose[9] = {8, 8, 7, 5, 4, 4, 3, 2, 2}, occ_advance[9] = {8, 8, 7, 5, 4, 3, 2, 2, 2};
  const int j = J < 1 ? 1 : (J > 8 ? 8 : J);
  const double t = 0.49 + 0.04 * j * j;
  const double overlap = j <= 5 ? 3.0 : 1.3;
  auto rounds = [overlap](double waves, int occ) {
    const double r = waves / (1024.0 * (occ < overlap ? occ : overlap));
    return r > 1.0 ? r : 1.0;
  };
  double us = rounds(B / 4.0, occ_advance[j]) * p.n[p.levels] * t;
  for (int l = 0; l < p.levels; ++l) {
    const double segs = (double)B * p.n[l + 1];
    us += rounds(segs / 2.0, occ_compose[j]) * (p.g[l] - 1) * 1.3 * t;
    us += rounds(segs / 4.0, occ_advance[j]) * (p.g[l] - 1) * t;
    us += 12.0;
  }
  return us;
}
// levels < 0: choose (multi-level only when the model promises at least 15 % over the walk); otherwise build
// the plan with that many levels of groups of g (a level is dropped when it would not leave two groups)
inline PrefixPlan plan_prefix(int nchunk, int levels = -1, int g = 0, int B = 1, int J = 8) {
  auto build = [&](int lv, int gg) {
    PrefixPlan p;
    p.levels = 0;
    p.n[0] = nchunk;
    for (int l = 0; l < 3; ++l) {
      const bool use = l < lv && gg >= 2 && p.n[l] >= 2 * gg;
      p.g[l] = use ? gg : 1;
      p.n[l + 1] = (p.n[l] + p.g[l] - 1) / p.g[l];
      if (use) p.levels = l + 1;
    }
    p.time_us = prefix_plan_time_us(p, B, J);
    return p;
  };
  if (levels >= 0) return build(levels > 3 ? 3 : levels, g);
  const PrefixPlan walk = build(0, 0);
  PrefixPlan best = walk;
  for (int lv = 1; lv <= 3; ++lv)
    for (int gg = 2; gg <= 64; ++gg) {
      const PrefixPlan p = build(lv, gg);
      if (p.levels == lv && p.time_us < best.time_us) best = p;
    }
  return best.time_us < 0.85 * walk.time_us ? best : walk;
}

// ---------------------------------------------------------------------------
// replay: the reference recurrence over samples [n0, n1) from a known state.
// Accumulates sum(log D_n) and sum(x_n^2 / D_n); flags the first D_n < 0 with
// n >= 1 (cholesky.h:176 -- sample 0 is never checked, :100-117).  When
// MATERIALIZE, writes the factor in the reference's storage (cholesky.h:76-78,
// :703-706): phi[:, n] (move n -> n+1), u[:, n-1] = U~(t_n), W[:, n], D[n].
// ---------------------------------------------------------------------------
template <int JR, int JC, int MATERIALIZE, bool FAST, class Src>
CLR_HD void replay_chunk(const Problem<JR, JC>& p, Src& src, int L, int N, int n0,
                         const double* start /* P[SZ] f[J] or nullptr => zero */,
                         double* logdet_out, double* quad_out, int* flag_out,
                         double* phi_o, double* u_o, double* W_o, double* D_o, long fstride,
                         double* end_out = nullptr /* state after the chunk: P[SZ] f[J] */,
                         int warm = 0 /* leading steps that only warm the state up (wave-uniform) */,
                         double* warm_out = nullptr /* state after them: P[SZ] f[J] */) {
  // MATERIALIZE: 0 = nothing is stored; 1 = the reference's storage (cholesky.h:76-78,
  // :703-706: phi[:, n], u[:, n-1], W[:, n], D[n] with element (j, n) at [j + J n];
  // pointers are the problem's arrays); 2 = chunk-interleaved device layout: the
  // pointers are already offset to this lane's chunk column and element (local step
  // i, row j) is at [(i * J + j) * fstride] -- the 64 lanes of a wave then store 512
  // contiguous bytes per instruction (u and phi both at their own sample's slot);
  // 3 = the LEAN chunk-interleaved layout: W and D as in 2, phi and u not stored at all
  // (consumers regenerate them from t and the coefficients: expand_lean_factor_kernel).
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  double P[SZ], f[J];
  if (start) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) P[i] = start[i];
    CLR_UNROLL
    for (int i = 0; i < J; ++i) f[i] = start[SZ + i];
  } else {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) P[i] = 0.0;
    CLR_UNROLL
    for (int i = 0; i < J; ++i) f[i] = 0.0;
  }

  double quad = 0.0;
  LogProduct lp;
  lp.init();
  int flag = 0;
  // Every lane runs L steps (wave-uniform, see DirectSeries); steps at or beyond
  // the end of the series (short last chunk, lanes past the last chunk) are
  // computed on padding and contribute nothing.
  src.prologue();
  double tn = src.t(0);
  double t_next = src.t(1);
  double diag_n = src.diag(0), y_n = src.y(0);
  for (int i = 0; i < L; ++i) {
    src.step_begin(i);
    const int n = n0 + i;
    const bool valid = n < N && i >= warm;
    if (warm_out && i == warm) {  // (wave-uniform branch, taken once: the warmed-up state, stored rather than kept)
      CLR_UNROLL
      for (int k = 0; k < SZ; ++k) warm_out[k] = P[k];
      CLR_UNROLL
      for (int k = 0; k < J; ++k) warm_out[SZ + k] = f[k];
    }
    const double t_cur_next = t_next, diag_cur = diag_n, y_cur = y_n;
    if (i + 1 < L) {
      t_next = src.t(i + 2);
      diag_n = src.diag(i + 1);
      y_n = src.y(i + 1);
    }

    double u[J], v[J];
    features_uv<JR, JC, FAST>(p, tn, u, v);

    double q[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc += P[sym(k, j)] * u[k];
      q[j] = acc;
    }
    double s = 0.0, uf = 0.0;
    CLR_UNROLL
    for (int j = 0; j < J; ++j) { s += u[j] * q[j]; uf += u[j] * f[j]; }
    const double D = p.diagonal(diag_cur) - s;
    // (the IEEE division also on the warm-started path: v_rcp + two Newton steps there measured no difference,
    //  2.23-2.25 against 2.21-2.25 ms same box, profiles/r03p_warm_chunks.txt)
    const double invD = 1.0 / D;
    const double x = y_cur - uf;
    if (valid) {
      if (n >= 1 && D < 0.0) flag = 1;  // cholesky.h:176 (sample 0 is never checked)
      lp.mul(D);
      quad += x * x * invD;
    }

    double z[J], W[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      z[j] = v[j] - q[j];
      W[j] = z[j] * invD;
    }
    if (MATERIALIZE == 1 && valid) {
      D_o[n] = D;
      CLR_UNROLL
      for (int j = 0; j < J; ++j) W_o[(long)J * n + j] = W[j];
      if (n >= 1) {
        CLR_UNROLL
        for (int j = 0; j < J; ++j) u_o[(long)J * (n - 1) + j] = u[j];
      }
    }
    if (MATERIALIZE == 2 && valid) {
      store_stream(D_o + (long)i * fstride, D);
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        store_stream(W_o + ((long)i * J + j) * fstride, W[j]);
        store_stream(u_o + ((long)i * J + j) * fstride, u[j]);
      }
    }
    if (MATERIALIZE == 3 && valid) {  // lean: W and D only -- phi and u are pure functions of (t, coefficients), cholesky.h:127-147
      store_stream(D_o + (long)i * fstride, D);
      CLR_UNROLL
      for (int j = 0; j < J; ++j) store_stream(W_o + ((long)i * J + j) * fstride, W[j]);
    }
    {
      double phid[nz(JR + JC)];
      features_phi_distinct<JR, JC>(p, t_cur_next - tn, phid);
      if (MATERIALIZE == 1 && n + 1 < N) {
        CLR_UNROLL
        for (int j = 0; j < J; ++j) phi_o[(long)J * n + j] = phid[phi_index<JR>(j)];
      }
      if (MATERIALIZE == 2 && n + 1 < N) {
        CLR_UNROLL
        for (int j = 0; j < J; ++j) store_stream(phi_o + ((long)i * J + j) * fstride, phid[phi_index<JR>(j)]);
      }
      CLR_UNROLL
      for (int j = 0; j < J; ++j) f[j] = phid[phi_index<JR>(j)] * (f[j] + W[j] * x);
      decay_rank1_update<JR, JC>(phid, z, W, P);
    }
    tn = t_cur_next;
    src.step_end(i);
  }
  *logdet_out = lp.log_value();
  *quad_out = quad;
  *flag_out = flag;
  if (end_out) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) end_out[i] = P[i];
    CLR_UNROLL
    for (int i = 0; i < J; ++i) end_out[SZ + i] = f[i];
  }
}

// -0.5 (quad + logdet + N log 2 pi) with the -inf rules of
// celerite/celerite.py:211-218.
CLR_HD double combine_loglike(double logdet, double quad, int N) {
  if (!isfinite(logdet)) return -INFINITY;
  const double ll = -0.5 * (quad + logdet + N * 1.8378770664093453 /* log(2 pi) */);
  return isfinite(ll) ? ll : -INFINITY;
}

}  // namespace clr
