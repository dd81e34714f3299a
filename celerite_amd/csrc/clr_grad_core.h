// celerite_amd/csrc/clr_grad_core.h -- CholeskySolver.grad_log_likelihood parallel in n (widths 1..8).
//
// The reference differentiates compute (cholesky.h:41-210) + dot_solve (:326-401) in forward mode
// (celerite/solver.cpp:347-463): one tangent recurrence per partial on top of the base recurrence.  A tangent
// recurrence is LINEAR in the tangent state (dS, df) once the base trajectory is fixed:
//     dS_{n+1} = F_n dS_n F_n^T + (terms explicit in the direction),   F_n = Phi_n (I - w_n u_n^T)
//     df_{n+1} = F_n df_n - F_n dS_n u_n x_n / D_n + (explicit terms)
// so over a chunk of samples [n0, n1) that starts from the TRUE base state (the start states the scan already
// produced, clr_batch_kernels.h) a tangent splits into
//   * the tangent from a ZERO tangent start, driven by the direction's explicit terms (grad_chunk: every
//     (chunk, direction) independently -- this is where the time goes), and
//   * the homogeneous propagation of the tangent state the chunk starts with, which needs only three riders of
//     the base trajectory, shared by all directions (grad_riders_chunk):
//         AA = F_{n1-1} ... F_{n0}          eta = sum_n r_n x_n / D_n        JJ = sum_n r_n r_n^T / D_n,
//         r_n = (F_{n-1} ... F_{n0})^T u_n:
//         dS_end = AA dS_0 AA^T            df_end = AA (df_0 - dS_0 eta)
//         d(log det) = -<JJ, dS_0>         d(quad) = -2 eta . df_0 + eta^T dS_0 eta
//     (the sum over n of the quad terms telescopes; derivation in DESIGN.md section 3.1).
// FORWARD mode: grad_combine walks the chunks of one (problem, direction): a 2 J^3 update per chunk, nothing per sample.
// REVERSE mode (the default, second half of this file): the transposed chunk maps carry the ADJOINT of the base state
// backwards over the chunks, and one reverse sweep per chunk yields all partials at once.
//
// Directions in the reference's order (solver.cpp:379-406): jitter | a_real | c_real | a_comp | b_comp | c_comp |
// d_comp.  A wave carries the TWO directions of one group -- {a_real_j, c_real_j}, {a_comp_j, b_comp_j},
// {c_comp_j, d_comp_j}, {jitter} -- on one base recurrence, with the group's term swapped to the LAST position of
// its kind when the coefficients and the start state are loaded: a direction only touches the rows of its own
// term (U~, V~, phi: cholesky.h:129-147), and with the swap those rows are compile-time constants (the recurrence
// does not care about the order of the terms; sums change in the last bits only).
#pragma once

#include "clr_core.h"

namespace clr {

template <int JR, int JC>
struct GradShape {
  static constexpr int J = JR + 2 * JC;
  static constexpr int SZ = J * (J + 1) / 2;
  static constexpr int NG = 1 + 2 * JR + 4 * JC;   // partials
  static constexpr int GROUPS = 1 + JR + 2 * JC;   // waves per 64 chunks
  static constexpr int OUT = SZ + J + 2;           // per (chunk, direction): G[SZ] g[J] d(log det) d(quad), zero tangent start
  static constexpr int RID = J * J + J + SZ;       // per chunk: AA[J][J] eta[J] JJ[SZ]
};

// Wave-level "some lane wants" / "this lane is the first active one" (host: the single lane itself).
#if defined(__HIP_DEVICE_COMPILE__)
#define CLR_WAVE_ANY(cond) (__any(cond) != 0)
#define CLR_FIRST_ACTIVE_LANE() ((int)threadIdx.x == __ffsll((unsigned long long)__ballot(1)) - 1)
#else
#define CLR_WAVE_ANY(cond) (cond)
#define CLR_FIRST_ACTIVE_LANE() (true)
#endif

// Where the reverse mode's forward pass stores base states for the sweep to continue from (clr_grad_core.h, second
// half).  Slots are handed out in step order, the same for all lanes of a wave; flag[i] says what happened before local
// step i (0 nothing, 1 state stored in the next slot, 2 wanted but no slot left) -- one byte per step and wave, written
// by the wave's first active lane, read back by the sweep.
//   K > 0:  every K steps.
//   K == 0: adaptive -- a state is stored as soon as, for SOME lane of the wave, the decay accumulated over the moves
//           since the last stored state would exceed the growth budget: scale x (t - t_stored) > 1, scale = (largest
//           decay rate of the problem) / CLR_GRAD_GROWTH_BUDGET.  Reconstruction errors grow like exp(2 c T): the
//           budget 3 bounds the growth over any stretch the sweep rebuilds and uses by 4e2, and over the stretch plus
//           the stored state's own move -- what the drift check at that state sees -- by 2e5.  Dense series store a
//           handful of states per chunk, sparse ones one every few samples, a lone long gap exactly where it is.
//   span > 1 (adaptive rule only; round 4): a state the rule asks for fewer than `span` steps after the last STORED one
//           is not stored but marked flag[i] = 16 + m (m = steps since that stored state): the sweep rebuilds it FORWARDS
//           from the stored state over the m recorded steps, S <- Phi (S + D w w^T) Phi, f <- Phi (f + w x) -- the
//           forward recurrence itself (a contraction: stable whatever the gaps), from w, D, x of the record and the
//           times.  A series that forgets between any two samples then stores every span-th state (44 / span instead
//           of 44 doubles per sample through HBM, twice) for (span - 1) / 2 rebuilt steps per step on average.
#define CLR_GRAD_GROWTH_BUDGET 3.0
#define CLR_GRAD_FLAG_REBUILD 16
struct GradStore {
  double* ck = nullptr;           // [slot][SZ + J], element k of a slot at ck[(slot (SZ + J) + k) rstride]
  unsigned char* flag = nullptr;  // [L]
  int K = 0;
  int nalloc = 0;                 // slots
  double* count = nullptr;        // (forward pass) out: slots this lane's chunk used; (sweep) in
  int span = 1;                   // (adaptive rule) stored states at least this many steps apart; 1: every one the rule asks for
};
template <int JR, int JC>
CLR_HD double grad_store_scale(const Problem<JR, JC>& p) {
  double cm = 0.0;
  CLR_UNROLL
  for (int j = 0; j < JR; ++j) cm = fmax(cm, fabs(p.cr[j]));
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) cm = fmax(cm, fabs(p.cc[j]));
  return cm / CLR_GRAD_GROWTH_BUDGET;
}

// group -> kind (0 jitter, 1 real term, 2 complex a/b, 3 complex c/d), term, the two direction numbers (-1: none)
template <int JR, int JC>
CLR_HD void grad_group(int g, int* kind, int* term, int* q0, int* q1) {
  if (g == 0) { *kind = 0; *term = 0; *q0 = 0; *q1 = -1; return; }
  g -= 1;
  if (g < JR) { *kind = 1; *term = g; *q0 = 1 + g; *q1 = 1 + JR + g; return; }
  g -= JR;
  if (g < JC) { *kind = 2; *term = g; *q0 = 1 + 2 * JR + g; *q1 = 1 + 2 * JR + JC + g; return; }
  g -= JC;
  *kind = 3; *term = g; *q0 = 1 + 2 * JR + 2 * JC + g; *q1 = 1 + 2 * JR + 3 * JC + g;
}

template <int J, int R>
CLR_HD void grad_add_col(const double* S, double c, double* y) {
  CLR_UNROLL
  for (int i = 0; i < J; ++i) y[i] = fma(S[sym(i, R)], c, y[i]);
}

// ---------------------------------------------------------------------------
// One chunk, one group: the base recurrence from `start` (packed S[SZ] f[J] in the problem's own row order, or
// null = zero state) and the two tangents of the group from zero tangent states.  Coefficients are read through
// the pointers (this problem's rows) with the group's term swapped to the end.  out0 / out1: [OUT] each, row order
// of the problem (out1 may be null: the jitter group has one direction).
// ---------------------------------------------------------------------------
template <int JR, int JC, bool FAST, class Src>
CLR_HD void grad_chunk(const double* a_real, const double* c_real, const double* a_comp, const double* b_comp,
                       const double* c_comp, const double* d_comp, double jitter, Src& src, int L, int N, int n0,
                       const double* start, int group, double* out0, double* out1) {
  using Sh = GradShape<JR, JC>;
  constexpr int J = Sh::J, SZ = Sh::SZ, M = JR + JC;
  constexpr int RR = JR > 0 ? JR - 1 : 0;  // the swapped real term's row
  constexpr int CR = JC > 0 ? J - 2 : 0;   // the swapped complex term's rows CR, CB = CR + 1
  constexpr int CB = JC > 0 ? J - 1 : 0;
  int kind, term, q0, q1;
  grad_group<JR, JC>(group, &kind, &term, &q0, &q1);

  // the problem with the group's term last, and the row map working row -> problem row
  Problem<JR, JC> p;
  int perm[J];
  {
    const int sr = kind == 1 ? term : JR - 1, sc = kind >= 2 ? term : JC - 1;
    p.sum_ar = 0.0;
    p.sum_ac = 0.0;
    CLR_UNROLL
    for (int j = 0; j < JR; ++j) {
      const int o = j == JR - 1 ? sr : (j == sr ? JR - 1 : j);
      p.ar[j] = a_real[o];
      p.cr[j] = c_real[o];
      perm[j] = o;
    }
    CLR_UNROLL
    for (int j = 0; j < JC; ++j) {
      const int o = j == JC - 1 ? sc : (j == sc ? JC - 1 : j);
      p.ac[j] = a_comp[o];
      p.bc[j] = b_comp[o];
      p.cc[j] = c_comp[o];
      p.dc[j] = d_comp[o];
      perm[JR + 2 * j] = JR + 2 * o;
      perm[JR + 2 * j + 1] = JR + 2 * o + 1;
    }
    // K(0) is summed in the problem's own order (cholesky.h:98), whatever the swap
    CLR_UNROLL
    for (int j = 0; j < JR; ++j) p.sum_ar += a_real[j];
    CLR_UNROLL
    for (int j = 0; j < JC; ++j) p.sum_ac += a_comp[j];
    p.jitter = jitter;
  }

  double S[SZ], f[J];
  CLR_UNROLL
  for (int j = 0; j < J; ++j) {
    CLR_UNROLL
    for (int k = 0; k <= j; ++k) {
      const int a = perm[k], b = perm[j];
      S[tri(k, j)] = start ? start[a <= b ? a + b * (b + 1) / 2 : b + a * (a + 1) / 2] : 0.0;
    }
    f[j] = start ? start[SZ + perm[j]] : 0.0;
  }
  double dS[2][SZ], df[2][J], dld[2] = {0.0, 0.0}, dqd[2] = {0.0, 0.0};
  CLR_UNROLL
  for (int e = 0; e < 2; ++e) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) dS[e][i] = 0.0;
    CLR_UNROLL
    for (int i = 0; i < J; ++i) df[e][i] = 0.0;
  }

  src.prologue();
  double tn = src.t(0);
  double t_next = src.t(1);
  double diag_n = src.diag(0), y_n = src.y(0);
  for (int i = 0; i < L; ++i) {
    src.step_begin(i);
    const int n = n0 + i;
    const bool valid = n < N;
    const double t_cur_next = t_next, diag_cur = diag_n, y_cur = y_n;
    if (i + 1 < L) {
      t_next = src.t(i + 2);
      diag_n = src.diag(i + 1);
      y_n = src.y(i + 1);
    }
    const double dt = t_cur_next - tn;

    // ---- base step up to w, x (cholesky.h:126-179, :384-398) ----
    double u[J], v[J];
    features_uv<JR, JC, FAST>(p, tn, u, v);
    double q[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc += S[sym(k, j)] * u[k];
      q[j] = acc;
    }
    double s = 0.0, uf = 0.0;
    CLR_UNROLL
    for (int j = 0; j < J; ++j) { s += u[j] * q[j]; uf += u[j] * f[j]; }
    const double D = p.diagonal(diag_cur) - s;
    const double invD = 1.0 / D;
    const double x = y_cur - uf;
    const double xs = x * invD;
    double z[J], w[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      z[j] = v[j] - q[j];
      w[j] = z[j] * invD;
    }
    double phid[nz(M)], pp[nz(M * (M + 1) / 2)];
    features_phi_distinct<JR, JC>(p, dt, phid);
    CLR_UNROLL
    for (int b = 0; b < M; ++b) {
      CLR_UNROLL
      for (int a = 0; a <= b; ++a) pp[tri(a, b)] = phid[a] * phid[b];
    }

    // ---- the two tangents: everything that reads the state BEFORE this step ----
    CLR_UNROLL
    for (int e = 0; e < 2; ++e) {
      // the direction's explicit terms: d K(0), dU~ / dV~ at the term's rows
      double da = 0.0, duA = 0.0, duB = 0.0, dvA = 0.0, dvB = 0.0;
      if (kind == 0) {
        da = 1.0;                                                        // jitter (cholesky.h:98)
      } else if (kind == 1) {
        if (e == 0) { da = 1.0; duA = 1.0; }                             // a_real: U~ = a (cholesky.h:131)
      } else if (kind == 2) {
        if (e == 0) { da = 1.0; duA = v[CR]; duB = v[CB]; }          // a_comp: U~ = (a cd + b sd, a sd - b cd)
        else { duA = v[CB]; duB = -v[CR]; }                          // b_comp
      } else if (e == 1) {                                               // d_comp: the phase d t
        duA = -tn * u[CB]; duB = tn * u[CR];
        dvA = -tn * v[CB]; dvB = tn * v[CR];
      }
      double dq[J];
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        double acc = 0.0;
        CLR_UNROLL
        for (int k = 0; k < J; ++k) acc += dS[e][sym(k, j)] * u[k];
        dq[j] = acc;
      }
      double duq = 0.0, duf = 0.0;
      if (kind == 1) {
        if (JR > 0) { grad_add_col<J, RR>(S, duA, dq); duq = duA * q[RR]; duf = duA * f[RR]; }
      } else if (kind >= 2) {
        if (JC > 0) {
          grad_add_col<J, CR>(S, duA, dq);
          grad_add_col<J, CB>(S, duB, dq);
          duq = duA * q[CR] + duB * q[CB];
          duf = duA * f[CR] + duB * f[CB];
        }
      }
      double ds = duq, dx = duf;
      CLR_UNROLL
      for (int j = 0; j < J; ++j) { ds += u[j] * dq[j]; dx += u[j] * df[e][j]; }
      dx = -dx;
      const double dD = da - ds;
      if (valid) {
        dld[e] = fma(dD, invD, dld[e]);
        dqd[e] += (2.0 * dx - xs * dD) * xs;
      }
      double dz[J], dw[J];
      CLR_UNROLL
      for (int j = 0; j < J; ++j) dz[j] = -dq[j];
      if (kind == 3 && JC > 0) { dz[CR] += dvA; dz[CB] += dvB; }
      CLR_UNROLL
      for (int j = 0; j < J; ++j) dw[j] = (dz[j] - w[j] * dD) * invD;
      // dS <- Phi (dS + dz w^T + z dw^T) Phi ; df <- Phi (df + dw x + w dx)   (the d Phi parts follow the base update)
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        CLR_UNROLL
        for (int k = 0; k <= j; ++k)
          dS[e][tri(k, j)] = pp[tri(phi_index<JR>(k), phi_index<JR>(j))] *
                             fma(dz[k], w[j], fma(z[k], dw[j], dS[e][tri(k, j)]));
        df[e][j] = phid[phi_index<JR>(j)] * (df[e][j] + fma(dw[j], x, w[j] * dx));
      }
    }

    // ---- base update ----
    CLR_UNROLL
    for (int j = 0; j < J; ++j) f[j] = phid[phi_index<JR>(j)] * (f[j] + w[j] * x);
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL
      for (int k = 0; k <= j; ++k)
        S[tri(k, j)] = pp[tri(phi_index<JR>(k), phi_index<JR>(j))] * fma(z[k], w[j], S[tri(k, j)]);
    }

    // ---- d Phi: phi = exp(-c dt) (cholesky.h:130,140) => d phi / dc = -dt phi on the term's rows, i.e.
    //      dS_ik += -dt ([i in term] + [k in term]) S_ik(new),  df_i += -dt [i in term] f_i(new) ----
    if (kind == 1) {
      if (JR > 0) {
        CLR_UNROLL
        for (int k = 0; k < J; ++k) dS[1][sym(RR, k)] = fma(k == RR ? -2.0 * dt : -dt, S[sym(RR, k)], dS[1][sym(RR, k)]);
        df[1][RR] = fma(-dt, f[RR], df[1][RR]);
      }
    } else if (kind == 3) {
      if (JC > 0) {
        CLR_UNROLL
        for (int k = 0; k < CR; ++k) {
          dS[0][sym(CR, k)] = fma(-dt, S[sym(CR, k)], dS[0][sym(CR, k)]);
          dS[0][sym(CB, k)] = fma(-dt, S[sym(CB, k)], dS[0][sym(CB, k)]);
        }
        dS[0][tri(CR, CR)] = fma(-2.0 * dt, S[tri(CR, CR)], dS[0][tri(CR, CR)]);
        dS[0][tri(CR, CB)] = fma(-2.0 * dt, S[tri(CR, CB)], dS[0][tri(CR, CB)]);
        dS[0][tri(CB, CB)] = fma(-2.0 * dt, S[tri(CB, CB)], dS[0][tri(CB, CB)]);
        df[0][CR] = fma(-dt, f[CR], df[0][CR]);
        df[0][CB] = fma(-dt, f[CB], df[0][CB]);
      }
    }
    tn = t_cur_next;
    src.step_end(i);
  }

  // ---- store in the problem's row order ----
  CLR_UNROLL
  for (int e = 0; e < 2; ++e) {
    double* o = e == 0 ? out0 : out1;
    if (!o) continue;
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL
      for (int k = 0; k <= j; ++k) {
        const int a = perm[k], b = perm[j];
        o[a <= b ? a + b * (b + 1) / 2 : b + a * (a + 1) / 2] = dS[e][tri(k, j)];
      }
      o[SZ + perm[j]] = df[e][j];
    }
    o[SZ + J] = dld[e];
    o[SZ + J + 1] = dqd[e];
  }
}

// ---------------------------------------------------------------------------
// The riders of one chunk along the base trajectory from `start`: out = AA[J][J] row-major | eta[J] | JJ[SZ].
// Only samples of the series (n < N) enter eta and JJ; AA of the last chunk is never used.
// ---------------------------------------------------------------------------
// rec (may be null): per sample of the chunk w[J], D, x -- element k of local step i at rec[(i (J + 2) + k) rstride] --
// and end_out = the base state after the chunk's last sample (S[SZ] f[J]): what the reverse sweep
// (grad_backward_chunk) starts from.  The move after the LAST sample of the series is taken with dt = 0.
// store (GradStore, ck may be null): base states BEFORE selected local steps, which bounds how far the reverse sweep
// reconstructs states.
// RIDERS = false: only the record (the riders then come from the scan's element, grad_riders_from_element).
template <int JR, int JC, bool FAST, class Src, bool RIDERS = true>
CLR_HD void grad_riders_chunk(const Problem<JR, JC>& p, Src& src, int L, int N, int n0, const double* start,
                              double* out, double* rec = nullptr, long rstride = 0, double* end_out = nullptr,
                              GradStore store = GradStore()) {
  using Sh = GradShape<JR, JC>;
  constexpr int J = Sh::J, SZ = Sh::SZ;
  double S[SZ], f[J], AA[RIDERS ? J * J : 1], eta[RIDERS ? J : 1], JJ[RIDERS ? SZ : 1];
  CLR_UNROLL
  for (int i = 0; i < SZ; ++i) S[i] = start ? start[i] : 0.0;
  CLR_UNROLL
  for (int i = 0; i < J; ++i) f[i] = start ? start[SZ + i] : 0.0;
  if (RIDERS) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) JJ[i] = 0.0;
    CLR_UNROLL
    for (int i = 0; i < J; ++i) {
      eta[i] = 0.0;
      CLR_UNROLL
      for (int j = 0; j < J; ++j) AA[i * J + j] = i == j ? 1.0 : 0.0;
    }
  }
  const double store_scale = grad_store_scale<JR, JC>(p);
  double store_acc = 0.0;
  int store_used = 0, store_last = -(1 << 20);  // (local step of the last stored state)
  src.prologue();
  double tn = src.t(0);
  double t_next = src.t(1);
  double diag_n = src.diag(0), y_n = src.y(0);
  for (int i = 0; i < L; ++i) {
    src.step_begin(i);
    const bool valid = n0 + i < N;
    const double t_cur_next = t_next, diag_cur = diag_n, y_cur = y_n;
    if (i + 1 < L) {
      t_next = src.t(i + 2);
      diag_n = src.diag(i + 1);
      y_n = src.y(i + 1);
    }
    if (valid) {  // (the state freezes behind the end of the series: end_out is the state after the last sample)
      if (store.ck) {
        const double sdt = n0 + i + 1 < N ? store_scale * (t_cur_next - tn) : 0.0;
        bool fire;
        if (store.K > 0) {
          fire = i > 0 && i % store.K == 0;
        } else {
          const bool want = i > 0 && store_acc + sdt > 1.0;
          fire = CLR_WAVE_ANY(want);
        }
        if (fire && store.K == 0 && i - store_last < store.span) {
          // rebuilt by the sweep from the stored state i - store_last steps back (wave-uniform, like `fire`)
          if (store.flag && CLR_FIRST_ACTIVE_LANE()) store.flag[i] = (unsigned char)(CLR_GRAD_FLAG_REBUILD + (i - store_last));
          store_acc = 0.0;
        } else if (fire) {
          const bool room = store_used < store.nalloc;
          store_last = i;
          if (room) {
            double* o = store.ck + (long)store_used * (SZ + J) * rstride;
            CLR_UNROLL
            for (int k = 0; k < SZ; ++k) store_stream(o + (long)k * rstride, S[k]);
            CLR_UNROLL
            for (int k = 0; k < J; ++k) store_stream(o + (long)(SZ + k) * rstride, f[k]);
            ++store_used;
          }
          if (store.flag && CLR_FIRST_ACTIVE_LANE()) store.flag[i] = room ? 1 : 2;
          store_acc = 0.0;  // (the stored state's own move does not count: nothing is rebuilt across it)
        } else {
          store_acc += sdt;
        }
      }
      double u[J], v[J];
      features_uv<JR, JC, FAST>(p, tn, u, v);
      double q[J];
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        double acc = 0.0;
        CLR_UNROLL
        for (int k = 0; k < J; ++k) acc += S[sym(k, j)] * u[k];
        q[j] = acc;
      }
      double s = 0.0, uf = 0.0;
      CLR_UNROLL
      for (int j = 0; j < J; ++j) { s += u[j] * q[j]; uf += u[j] * f[j]; }
      const double D = p.diagonal(diag_cur) - s;
      const double invD = 1.0 / D;
      const double x = y_cur - uf;
      double z[J], w[J];
      CLR_UNROLL
      for (int j = 0; j < J; ++j) {
        z[j] = v[j] - q[j];
        w[j] = z[j] * invD;
      }
      if (rec) {  // (written once, read once by the reverse sweep: streaming stores)
        CLR_UNROLL
        for (int j = 0; j < J; ++j) store_stream(rec + ((long)i * (J + 2) + j) * rstride, w[j]);
        store_stream(rec + ((long)i * (J + 2) + J) * rstride, D);
        store_stream(rec + ((long)i * (J + 2) + J + 1) * rstride, x);
      }
      double r[J];
      if (RIDERS) {
        CLR_UNROLL
        for (int j = 0; j < J; ++j) {
          double acc = 0.0;
          CLR_UNROLL
          for (int k = 0; k < J; ++k) acc += AA[k * J + j] * u[k];
          r[j] = acc;
        }
        const double xs = x * invD;
        CLR_UNROLL
        for (int j = 0; j < J; ++j) {
          eta[j] = fma(r[j], xs, eta[j]);
          const double rj = r[j] * invD;
          CLR_UNROLL
          for (int k = 0; k <= j; ++k) JJ[tri(k, j)] = fma(r[k], rj, JJ[tri(k, j)]);
        }
      }
      double phid[nz(JR + JC)];
      features_phi_distinct<JR, JC>(p, n0 + i + 1 < N ? t_cur_next - tn : 0.0, phid);
      CLR_UNROLL
      for (int k = 0; k < J; ++k) {
        const double ph = phid[phi_index<JR>(k)];
        if (RIDERS) {
          CLR_UNROLL
          for (int j = 0; j < J; ++j) AA[k * J + j] = ph * fma(-w[k], r[j], AA[k * J + j]);
        }
        f[k] = ph * (f[k] + w[k] * x);
      }
      decay_rank1_update<JR, JC>(phid, z, w, S);
    }
    tn = t_cur_next;
    src.step_end(i);
  }
  if (end_out) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) end_out[i] = S[i];
    CLR_UNROLL
    for (int i = 0; i < J; ++i) end_out[SZ + i] = f[i];
  }
  if (store.count) *store.count = (double)store_used;
  if (RIDERS) {
    CLR_UNROLL
    for (int i = 0; i < J * J; ++i) out[i] = AA[i];
    CLR_UNROLL
    for (int i = 0; i < J; ++i) out[J * J + i] = eta[i];
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) out[J * J + J + i] = JJ[i];
  }
}

// ---------------------------------------------------------------------------
// The same three riders from the scan's OWN element of the chunk (A, b, C, eta_e, Jm: summarize_chunk, clr_core.h)
// and the chunk's start state (P, f) -- they are the derivatives of the element's maps at that state (chunk_update:
// S_end = C + A G A^T, f_end = b + A g, log det += log det(I + P Jm), quad += 2 eta.f - f^T Jm f + w^T G w with
// Mi = (I + P Jm)^-1, G = Mi P, g = Mi (f + P eta), w = Jm f - eta).  With dG = Mi dP Mi^T:
//     AA = A Mi          eta = Mi^T (Jm f - eta_e)          JJ = -Jm Mi   (symmetric: Jm Mi = (I + Jm P)^-1 Jm)
// One J x J Gauss-Jordan with partial pivoting per chunk instead of ~190 FMAs per SAMPLE.  The reverse sweep checks
// the result: the adjoint it arrives at for the chunk's first sample must equal the one the walk over these riders
// predicted (grad_backward_chunk: adj0_out).  start == nullptr: the zero state (Mi = I).
// ---------------------------------------------------------------------------
template <int J>
CLR_HD void grad_riders_from_element(const double* elem, const double* start, double* out) {
  constexpr int SZ = J * (J + 1) / 2;
  const double* A = elem;
  const double* eta_e = elem + J * J + J + SZ;
  const double* Jm = eta_e + J;
  double T[J][2 * J];  // [ I + P Jm | I ] -> [ I | Mi ]
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = (i == j) ? 1.0 : 0.0;
      if (start) {
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) acc = fma(start[sym(i, k)], Jm[sym(k, j)], acc);
      }
      T[i][j] = acc;
      T[i][J + j] = (i == j) ? 1.0 : 0.0;
    }
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
    for (int c = col; c < 2 * J; ++c) {
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
    for (int c = col + 1; c < 2 * J; ++c) T[col][c] *= inv;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      if (i == col) continue;
      const double m = T[i][col];
      CLR_UNROLL_J
      for (int c = col + 1; c < 2 * J; ++c) T[i][c] -= m * T[col][c];
    }
  }
  // AA = A Mi
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc = fma(A[i * J + k], T[k][J + j], acc);
      out[i * J + j] = acc;
    }
  }
  // eta = Mi^T (Jm f - eta_e)
  double wv[J];
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double acc = -eta_e[i];
    if (start) {
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) acc = fma(Jm[sym(i, k)], start[SZ + k], acc);
    }
    wv[i] = acc;
  }
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) {
    double acc = 0.0;
    CLR_UNROLL_J
    for (int k = 0; k < J; ++k) acc = fma(T[k][J + i], wv[k], acc);
    out[J * J + i] = acc;
  }
  // JJ = -sym(Jm Mi)
  CLR_UNROLL_J
  for (int j = 0; j < J; ++j) {
    CLR_UNROLL_J
    for (int i = 0; i <= j; ++i) {
      double a1 = 0.0, a2 = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) {
        a1 = fma(Jm[sym(i, k)], T[k][J + j], a1);
        a2 = fma(Jm[sym(j, k)], T[k][J + i], a2);
      }
      out[J * J + J + tri(i, j)] = -0.5 * (a1 + a2);
    }
  }
}

// ---------------------------------------------------------------------------
// One (problem, direction): walk the chunks.  riders: [nchunk][RID]; gout: this direction's record of chunk c at
// gout + c * stride ([OUT]).  Returns d(log det), d(quad) of the whole series.
// ---------------------------------------------------------------------------
template <int J>
CLR_HD void grad_combine(int nchunk, const double* riders, const double* gout, long stride, double* dld_out,
                         double* dquad_out) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ;
  double dS[SZ], df[J];
  CLR_UNROLL_J
  for (int i = 0; i < SZ; ++i) dS[i] = 0.0;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) df[i] = 0.0;
  double dld = 0.0, dqd = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    const double* R = riders + (long)c * RID;
    const double* G = gout + (long)c * stride;
    const double *AA = R, *eta = R + J * J, *JJ = R + J * J + J;
    double acc = 0.0;
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL_J
      for (int k = 0; k <= j; ++k) acc = fma(k == j ? 1.0 : 2.0, JJ[tri(k, j)] * dS[tri(k, j)], acc);
    }
    dld += G[SZ + J] - acc;
    double tmp[J];
    double e1 = 0.0, e2 = 0.0;
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {
      double a = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) a = fma(dS[sym(i, k)], eta[k], a);
      tmp[i] = a;
      e1 = fma(eta[i], df[i], e1);
      e2 = fma(eta[i], a, e2);
    }
    dqd += G[SZ + J + 1] - 2.0 * e1 + e2;
    if (c + 1 < nchunk) {
      double h[J], T[J * J];
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) h[i] = df[i] - tmp[i];
      CLR_UNROLL_J
      for (int i = 0; i < J; ++i) {
        double a = G[SZ + i];
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) a = fma(AA[i * J + k], h[k], a);
        df[i] = a;
        CLR_UNROLL_J
        for (int j = 0; j < J; ++j) {
          double b = 0.0;
          CLR_UNROLL_J
          for (int k = 0; k < J; ++k) b = fma(AA[i * J + k], dS[sym(k, j)], b);
          T[i * J + j] = b;
        }
      }
      CLR_UNROLL_J
      for (int j = 0; j < J; ++j) {
        CLR_UNROLL_J
        for (int i = 0; i <= j; ++i) {
          double b = G[tri(i, j)];
          CLR_UNROLL_J
          for (int k = 0; k < J; ++k) b = fma(T[i * J + k], AA[j * J + k], b);
          dS[tri(i, j)] = b;
        }
      }
    }
  }
  *dld_out = dld;
  *dquad_out = dqd;
}

// ---------------------------------------------------------------------------
// REVERSE mode.  With L = sum_n (log D_n + x_n^2 / D_n) and the chunk maps above, the adjoint of the base state at a
// chunk's first sample follows from the adjoint at its end (= the first sample of the next chunk) through the same
// three riders:
//     Sbar_0 = AA^T Sbar_e AA - sym((AA^T fbar_e) eta^T) - JJ + eta eta^T        fbar_0 = AA^T fbar_e - 2 eta
// (the transpose of the tangent maps; sym(a b^T) = (a b^T + b a^T) / 2; adjoints of symmetric matrices are kept
// symmetric, dL = sum over ALL i, k of Sbar_ik dS_ik).  grad_adjoint_walk: one problem, backwards over the chunks,
// adj[c] = the adjoint at the END of chunk c (Sbar[SZ] fbar[J]); the last chunk ends with zeros.
// ---------------------------------------------------------------------------
template <int J>
CLR_HD void grad_adjoint_walk(int nchunk, const double* riders, double* adj) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ, ADJ = SZ + J;
  double Sb[SZ], fb[J];
  CLR_UNROLL_J
  for (int i = 0; i < SZ; ++i) Sb[i] = 0.0;
  CLR_UNROLL_J
  for (int i = 0; i < J; ++i) fb[i] = 0.0;
  for (int c = nchunk - 1; c >= 0; --c) {
    double* o = adj + (long)c * ADJ;
    CLR_UNROLL_J
    for (int i = 0; i < SZ; ++i) o[i] = Sb[i];
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) o[SZ + i] = fb[i];
    if (c == 0) break;
    const double* R = riders + (long)c * RID;
    const double *AA = R, *eta = R + J * J, *JJ = R + J * J + J;
    double T[J * J], g[J];
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) {    // T = Sbar AA ; g = AA^T fbar
      double a = 0.0;
      CLR_UNROLL_J
      for (int k = 0; k < J; ++k) a = fma(AA[k * J + i], fb[k], a);
      g[i] = a;
      CLR_UNROLL_J
      for (int j = 0; j < J; ++j) {
        double b = 0.0;
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) b = fma(Sb[sym(i, k)], AA[k * J + j], b);
        T[i * J + j] = b;
      }
    }
    CLR_UNROLL_J
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL_J
      for (int i = 0; i <= j; ++i) {
        double b = eta[i] * eta[j] - JJ[tri(i, j)] - 0.5 * (g[i] * eta[j] + eta[i] * g[j]);
        CLR_UNROLL_J
        for (int k = 0; k < J; ++k) b = fma(AA[k * J + i], T[k * J + j], b);
        Sb[tri(i, j)] = b;
      }
    }
    CLR_UNROLL_J
    for (int i = 0; i < J; ++i) fb[i] = g[i] - 2.0 * eta[i];
  }
}

// ---------------------------------------------------------------------------
// One chunk backwards: from the base state after its last sample (end_state, as grad_riders_chunk left it) and the
// adjoint there (end_adj), over the stored w, D, x of its samples.  Per step the state BEFORE the step is
// reconstructed from the state after it,
//     G = S' / (phi phi^T)     S = G - D w w^T        h = f' / phi     f = h - w x
// and the adjoints follow cholesky.h:154-178 / :384-398 backwards (q = S u, D = a - u.q, z = v - q, w = z / D,
// x = y - u.f, S' = Phi (S + z w^T) Phi, f' = Phi (f + w x)):
//     Gbar = phi phi^T o Sbar'   hbar = phi o fbar'   m = Gbar w
//     wbar = hbar x + D m        xbar = hbar.w + 2 x / D          zbar = m + wbar / D
//     Dbar = 1 / D - x^2 / D^2 - wbar.w / D            qbar = -zbar - Dbar u          vbar = zbar
//     ubar = -xbar f - Dbar q + S qbar   (q = v - D w)           fbar = hbar - xbar u
//     Sbar = Gbar + sym(qbar u^T)
// every coefficient's partial is accumulated on the way: K(0) (jitter, a_real, a_comp) through Dbar; a_real through
// ubar; a_comp, b_comp, d_comp through ubar, vbar and the phase; c through d phi / dc = -dt phi:
//     -dt (2 sum_k Sbar'_jk S'_jk + fbar'_j f'_j) summed over the term's rows j.
// out[NG]: this chunk's share of dL / d(coefficient) (the caller sums the chunks and applies -1/2); adj0_out
// (may be null): the adjoint at the chunk's first sample, equal to what grad_adjoint_walk found for the end of the
// previous chunk -- the built-in consistency check of the tests.
// The reconstruction inverts a contraction: its rounding errors grow like exp(2 c T) over a time span T (measured on
// the host instantiation: exact to 1e-13 over c T = 8, lost over c T = 40).  So the forward pass stores states
// (GradStore: wherever the accumulated decay since the last stored one reaches the growth budget), the sweep continues
// from the stored state there -- and from the scan's start state (`start`, null = the zero state) at the chunk's first
// sample -- and reports how far its own reconstruction had drifted (mismatch_out: the certificate -- a problem whose
// drift exceeds the tolerance is handed to the forward-mode kernels above).
// ---------------------------------------------------------------------------
template <int JR, int JC, bool FAST, class Src>
CLR_HD void grad_backward_chunk(const Problem<JR, JC>& p, Src& src, int L, int N, int n0, const double* end_state,
                                const double* end_adj, const double* rec, long rstride, double* out,
                                double* adj0_out = nullptr, GradStore store = GradStore(),
                                double* mismatch_out = nullptr, const double* start = nullptr) {
  using Sh = GradShape<JR, JC>;
  constexpr int J = Sh::J, SZ = Sh::SZ, M = JR + JC, NG = Sh::NG;
  double S[SZ], f[J], Sb[SZ], fb[J];
  CLR_UNROLL
  for (int i = 0; i < SZ; ++i) { S[i] = end_state[i]; Sb[i] = end_adj[i]; }
  CLR_UNROLL
  for (int i = 0; i < J; ++i) { f[i] = end_state[SZ + i]; fb[i] = end_adj[SZ + i]; }
  double g_k0 = 0.0;                       // sum of Dbar: jitter, and every a_real / a_comp
  double g_ar[nz(JR)], g_cr[nz(JR)], g_ac[nz(JC)], g_bc[nz(JC)], g_cc[nz(JC)], g_dc[nz(JC)];
  CLR_UNROLL
  for (int j = 0; j < JR; ++j) { g_ar[j] = 0.0; g_cr[j] = 0.0; }
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) { g_ac[j] = 0.0; g_bc[j] = 0.0; g_cc[j] = 0.0; g_dc[j] = 0.0; }
  const int last = (N - n0 < L ? N - n0 : L) - 1;  // local index of the chunk's last sample
  double drift = 0.0;
  const double store_scale = grad_store_scale<JR, JC>(p);
  int store_left = store.count ? (int)*store.count : 0;  // slots this chunk's forward pass filled, taken from the top
  // the record and the times of a step are fetched one step ahead (one wave per SIMD: nothing else hides the latency)
  double w_n[J], D_n = 1.0, x_n = 0.0, t_n = 0.0, t_n1 = 0.0;
  auto fetch = [&](int i) {
    CLR_UNROLL
    for (int j = 0; j < J; ++j) w_n[j] = rec[((long)i * (J + 2) + j) * rstride];
    D_n = rec[((long)i * (J + 2) + J) * rstride];
    x_n = rec[((long)i * (J + 2) + J + 1) * rstride];
    t_n1 = t_n;          // (going backwards: the time after step i is the time of the step fetched before)
    t_n = src.t(i);
  };
  if (last >= 0) {
    t_n = src.t(last + 1);
    fetch(last);
  }
  for (int i = last; i >= 0; --i) {
    const int n = n0 + i;
    const double tn = t_n;
    const double dt = n + 1 < N ? t_n1 - tn : 0.0;
    double w[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) w[j] = w_n[j];
    const double D = D_n, x = x_n;
    if (i > 0) fetch(i - 1);
    const double invD = recip_fast(D);
    double u[J], v[J], phid[nz(M)], iphid[nz(M)], pp[nz(M * (M + 1) / 2)], ipp[nz(M * (M + 1) / 2)];
    features_uv<JR, JC, FAST>(p, tn, u, v);
    features_phi_distinct<JR, JC>(p, dt, phid);
    CLR_UNROLL
    for (int a = 0; a < M; ++a) iphid[a] = recip_fast(phid[a]);
    CLR_UNROLL
    for (int bb = 0; bb < M; ++bb) {
      CLR_UNROLL
      for (int a = 0; a <= bb; ++a) {
        pp[tri(a, bb)] = phid[a] * phid[bb];
        ipp[tri(a, bb)] = iphid[a] * iphid[bb];
      }
    }
    // d phi / dc from the state AFTER the step, then the state before it
    double crow[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc = fma(Sb[sym(j, k)], S[sym(j, k)], acc);
      crow[j] = fma(fb[j], f[j], 2.0 * acc);
    }
    CLR_UNROLL
    for (int j = 0; j < JR; ++j) g_cr[j] = fma(-dt, crow[j], g_cr[j]);
    CLR_UNROLL
    for (int j = 0; j < JC; ++j) g_cc[j] = fma(-dt, crow[JR + 2 * j] + crow[JR + 2 * j + 1], g_cc[j]);
    // Gbar = phi phi^T o Sbar' (in place: Sb holds Gbar until the end of the step); S, f before the step
    double hb[J], mDw[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) mDw[j] = -D * w[j];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      CLR_UNROLL
      for (int k = 0; k <= j; ++k) {
        const int e = tri(phi_index<JR>(k), phi_index<JR>(j));
        Sb[tri(k, j)] = pp[e] * Sb[tri(k, j)];
        S[tri(k, j)] = fma(mDw[k], w[j], ipp[e] * S[tri(k, j)]);
      }
      hb[j] = phid[phi_index<JR>(j)] * fb[j];
      f[j] = fma(-w[j], x, iphid[phi_index<JR>(j)] * f[j]);
    }
    const int stored = (store.ck && store.flag && i > 0) ? store.flag[i] : 0;
    if (stored == 2) drift = INFINITY;  // (the forward pass ran out of slots: the rebuilt states are not certified)
    if (stored == 1 || i == 0) {
      // a state known independently -- stored by the forward pass, or (i = 0) the chunk's start state from the scan:
      // measure how far the reconstruction has drifted since the last one, then continue from the known state.
      // (A stored state whose own move is a long one -- beyond the growth budget by itself -- is not measured: the
      //  one reconstruction across that move is never used; the adjoint mismatch still covers the chunk.)
      if (stored == 1) --store_left;
      const double* o = i > 0 ? store.ck + (long)store_left * (SZ + J) * rstride : start;
      const long os = i > 0 ? rstride : 1;
      const bool measure = store.K > 0 || !(store_scale * dt > 1.0);
      // scale of the comparison: the state before the step is the DIFFERENCE S = G - D w w^T, f = h - w x; where the
      // series has forgotten its past S is tiny next to the terms it is the difference of, and it is their rounding
      // (and the dynamics' natural scale), not S's own size, that an error must be held against
      double wmax = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) wmax = fmax(wmax, fabs(w[k]));
      double big = fabs(D) * wmax * wmax, dev = 0.0;
      CLR_UNROLL
      for (int k = 0; k < SZ; ++k) {
        const double e = o ? o[(long)k * os] : 0.0;
        big = fmax(big, fabs(e));
        dev = fmax(dev, fabs(e - S[k]));  // (fmax drops a NaN; a NaN reconstruction shows in the partials instead)
        S[k] = e;
      }
      double bigf = wmax * fabs(x), devf = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) {
        const double e = o ? o[(long)(SZ + k) * os] : 0.0;
        bigf = fmax(bigf, fabs(e));
        devf = fmax(devf, fabs(e - f[k]));
        f[k] = e;
      }
      if (o && measure) {  // (the zero state of the first chunk: an absolute deviation has no scale to compare with)
        double r = big > 0.0 ? dev / big : (dev == 0.0 ? 0.0 : INFINITY);
        if (bigf > 0.0) r = fmax(r, devf / bigf);
        if (!(r <= drift)) drift = r;
      }
    }
    if (stored >= CLR_GRAD_FLAG_REBUILD) {
      // the state before this step, forwards from the last stored state (GradStore: span): steps i - nb .. i - 1
      const int nb = stored - CLR_GRAD_FLAG_REBUILD;
      const double* o = store.ck + (long)(store_left - 1) * (SZ + J) * rstride;
      CLR_UNROLL
      for (int k = 0; k < SZ; ++k) S[k] = o[(long)k * rstride];
      CLR_UNROLL
      for (int k = 0; k < J; ++k) f[k] = o[(long)(SZ + k) * rstride];
      double tp = src.t(i - nb);
      for (int s = i - nb; s < i; ++s) {
        const double tq = src.t(s + 1);
        double ph[nz(M)], ww[J], zz[J];
        features_phi_distinct<JR, JC>(p, tq - tp, ph);
        tp = tq;
        const double Ds = rec[((long)s * (J + 2) + J) * rstride], xs_ = rec[((long)s * (J + 2) + J + 1) * rstride];
        CLR_UNROLL
        for (int j = 0; j < J; ++j) {
          ww[j] = rec[((long)s * (J + 2) + j) * rstride];
          zz[j] = Ds * ww[j];
        }
        CLR_UNROLL
        for (int k = 0; k < J; ++k) f[k] = ph[phi_index<JR>(k)] * fma(ww[k], xs_, f[k]);
        decay_rank1_update<JR, JC>(ph, zz, ww, S);
      }
    }
    // adjoints of the step
    double m[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc = fma(Sb[sym(j, k)], w[k], acc);
      m[j] = acc;
    }
    double wb[J], zb[J], xbar = 2.0 * x * invD, wbw = 0.0;
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      wb[j] = fma(hb[j], x, D * m[j]);
      xbar = fma(hb[j], w[j], xbar);
      zb[j] = fma(wb[j], invD, m[j]);
      wbw = fma(wb[j], w[j], wbw);
    }
    const double Dbar = invD - (x * invD) * (x * invD) - wbw * invD;
    g_k0 += Dbar;
    double qb[J], qh[J], ub[J];
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      qb[j] = -zb[j] - Dbar * u[j];
      qh[j] = 0.5 * qb[j];
    }
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      const double q = fma(mDw[j], 1.0, v[j]);  // q = v - z, z = D w
      double acc = -xbar * f[j] - Dbar * q;
      CLR_UNROLL
      for (int k = 0; k < J; ++k) acc = fma(S[sym(j, k)], qb[k], acc);
      ub[j] = acc;
    }
    CLR_UNROLL
    for (int j = 0; j < J; ++j) {
      fb[j] = fma(-xbar, u[j], hb[j]);
      CLR_UNROLL
      for (int k = 0; k <= j; ++k) Sb[tri(k, j)] = fma(qh[k], u[j], fma(u[k], qh[j], Sb[tri(k, j)]));
    }
    // coefficients: U~ = a (real), (a cd + b sd, a sd - b cd) and V~ = (cd, sd) (complex), cholesky.h:129-147
    CLR_UNROLL
    for (int j = 0; j < JR; ++j) g_ar[j] += ub[j];
    CLR_UNROLL
    for (int j = 0; j < JC; ++j) {
      const int k = JR + 2 * j;
      const double cd = v[k], sd = v[k + 1];
      g_ac[j] += fma(ub[k], cd, ub[k + 1] * sd);
      g_bc[j] += fma(ub[k], sd, -ub[k + 1] * cd);
      g_dc[j] = fma(tn, (ub[k + 1] * u[k] - ub[k] * u[k + 1]) + (zb[k + 1] * cd - zb[k] * sd), g_dc[j]);
    }
  }
  out[0] = g_k0;
  CLR_UNROLL
  for (int j = 0; j < JR; ++j) { out[1 + j] = g_ar[j] + g_k0; out[1 + JR + j] = g_cr[j]; }
  CLR_UNROLL
  for (int j = 0; j < JC; ++j) {
    out[1 + 2 * JR + j] = g_ac[j] + g_k0;
    out[1 + 2 * JR + JC + j] = g_bc[j];
    out[1 + 2 * JR + 2 * JC + j] = g_cc[j];
    out[1 + 2 * JR + 3 * JC + j] = g_dc[j];
  }
  (void)NG;
  if (mismatch_out) *mismatch_out = drift;
  if (adj0_out) {
    CLR_UNROLL
    for (int i = 0; i < SZ; ++i) adj0_out[i] = Sb[i];
    CLR_UNROLL
    for (int i = 0; i < J; ++i) adj0_out[SZ + i] = fb[i];
  }
}

}  // namespace clr
