// celerite_amd/csrc/grad_kernels.hip -- CholeskySolver.grad_log_likelihood on the GPU.
//
// The reference differentiates by running its solver template on Eigen's forward-mode
// AutoDiffScalar (celerite/solver.cpp:347-463): compute (cholesky.h:41-210) and
// dot_solve (:326-401) on dual numbers carrying 1 + 2 J_real + 4 J_comp partials.
// Forward mode is one independent tangent recurrence per partial on top of the same
// base recurrence, so the natural GPU mapping is ONE WAVE PER PARTIAL (grid.x = number
// of partials): every wave recomputes the base step (the wave-per-problem layout of
// wide_kernels.hip: S distributed over the lanes, DPP reductions) and carries the
// tangent of its own direction p:
//     q = S u                      dq = dS u + S du
//     D = a - u.q                  dD = da - (du.q + u.dq)
//     z = v - q ; w = z / D        dz = dv - dq ; dw = (dz - w dD) / D
//     x = y - u.f                  dx = -(du.f + u.df)
//     S <- Phi (S + z w^T) Phi     dS <- the product rule on phi_i (phi_k S_ik + z_i phi_k w_k)
//     f <- Phi (f + w x)           df <- dPhi (f + w x) + Phi (df + dw x + w dx)
//     log det += log D             d(log det) += dD / D
//     quad += x^2 / D              d(quad) += (2 x dx - x^2 dD / D) / D
// A direction only touches the rows of its own term: per-lane constants select
// du = du0 + duc cos + dus sin + ddf t (us cos - uc sin), dphi = -dcf dx phi, and da = 1
// for jitter / a_real / a_comp (K(0) = sum of the amplitudes + jitter, cholesky.h:98).
// General terms (A, U, V: cholesky.h:65-72,114-116,148-152) are constants of the
// differentiation and enter as extra rows whose u, v are read from memory two steps
// ahead.  Result: value = -(quad + log det + pi log N) / 2 -- the reference's constant,
// solver.cpp:415 -- and grad[p] = -(d quad + d log det) / 2.
#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_wide.h"

namespace clr {

namespace {

template <int WMAX, bool FAST, bool CHUNKED = false>
__global__ void __launch_bounds__(64) wide_grad_kernel(const GradParams Pin) {
  GradParams P = Pin;
  if (Pin.B > 0) {  // batched: shift every pointer to problem blockIdx.y (wave-uniform)
    const long b = blockIdx.y;
    if (!CHUNKED && Pin.only_level && Pin.only_level[b] < 2) return;  // (settled by the chunk-parallel gradient)
    if (CHUNKED && Pin.only_level && Pin.only_level[b] >= 2) return;  // (left to the sequential form)
    const int NG = 1 + 2 * Pin.J_real + 4 * Pin.J_comp;
    if (P.A) P.A += b * Pin.A_stride;   // (general terms per problem; strides 0: shared blocks)
    if (P.U) { P.U += b * Pin.U_stride; P.V += b * Pin.V_stride; }
    P.a_real += b * Pin.J_real; P.c_real += b * Pin.J_real;
    P.a_comp += b * Pin.J_comp; P.b_comp += b * Pin.J_comp; P.c_comp += b * Pin.J_comp; P.d_comp += b * Pin.J_comp;
    P.jitter = Pin.jitter_b[b];
    P.t += b * Pin.t_stride; P.diag += b * Pin.diag_stride; P.y += b * Pin.y_stride;
    P.out_value += b; P.out_grad += b * NG; P.out_status += b;
  }
  using G = WideGeom<WMAX>;
  constexpr int LPR = G::LPR, COLS = G::COLS;
  __shared__ __attribute__((aligned(16))) double ubuf[2][WMAX], dubuf[2][WMAX];
  __shared__ __attribute__((aligned(16))) double pbuf[2][WMAX], dpbuf[2][WMAX];
  __shared__ __attribute__((aligned(16))) double wbuf[WMAX], dwbuf[WMAX];
  const int lane = threadIdx.x;
  const int row = lane / LPR, seg = lane % LPR;
  const bool writer = seg == 0;
  const int JR = P.J_real, JC = P.J_comp, JG = P.J_general, N = P.N;
  const int Wc = JR + 2 * JC;

  // ---- this lane's row ------------------------------------------------------------
  double u0 = 0.0, uc = 0.0, us = 0.0, v0 = 0.0, vc = 0.0, vs = 0.0, cdec = 0.0, dfreq = 0.0;
  int pair = -1;        // complex pair index of this row, if any
  bool cosrow = false;
  const double *ug = nullptr, *vg = nullptr;  // general row: U[j][:], V[j][:]
  if (row < JR) {
    u0 = P.a_real[row];
    v0 = 1.0;
    cdec = P.c_real[row];
  } else if (row < Wc) {
    pair = (row - JR) >> 1;
    cosrow = ((row - JR) & 1) == 0;
    const double a = P.a_comp[pair], b = P.b_comp[pair];
    if (cosrow) { uc = a; us = b; vc = 1.0; }   // cholesky.h:143,145
    else        { uc = -b; us = a; vs = 1.0; }  // cholesky.h:144,146
    cdec = P.c_comp[pair];
    dfreq = P.d_comp[pair];
  } else if (row < Wc + JG) {
    ug = P.U + (long)(row - Wc) * N;
    vg = P.V + (long)(row - Wc) * N;
  }
  // ---- this wave's direction (solver.cpp:379-406: jitter, a_real, c_real, a_comp,
  //      b_comp, c_comp, d_comp) ----------------------------------------------------
  double du0 = 0.0, duc = 0.0, dus = 0.0, dcf = 0.0, ddf = 0.0, da = 0.0;
  {
    int q = blockIdx.x;
    if (q == 0) {
      da = 1.0;
    } else if ((q -= 1) < JR) {
      da = 1.0;
      if (row == q) du0 = 1.0;
    } else if ((q -= JR) < JR) {
      if (row == q) dcf = 1.0;
    } else if ((q -= JR) < JC) {
      da = 1.0;
      if (pair == q) { if (cosrow) duc = 1.0; else dus = 1.0; }
    } else if ((q -= JC) < JC) {
      if (pair == q) { if (cosrow) dus = 1.0; else duc = -1.0; }
    } else if ((q -= JC) < JC) {
      if (pair == q) dcf = 1.0;
    } else {
      q -= JC;
      if (pair == q) ddf = 1.0;
    }
  }
  double sum_ar = 0.0, sum_ac = 0.0;  // cholesky.h:98
  for (int j = 0; j < JR; ++j) sum_ar += P.a_real[j];
  for (int j = 0; j < JC; ++j) sum_ac += P.a_comp[j];
  const double jitter = P.jitter;
  const bool has_general = P.A != nullptr;

  auto features = [&](double t, double dx, double ugen, double vgen, double* u, double* du,
                      double* v, double* dv, double* phi, double* dphi) {
    double sd, cs;
    sincos_phase<FAST>(dfreq * t, &sd, &cs);
    const double x = -cdec * dx;
    const double e = CLR_WAVE_ALL(fabs(x) < 0.0078125) ? exp_small(x) : exp(x);
    *phi = e;
    *dphi = -(dcf * dx) * e;
    *u = fma(uc, cs, fma(us, sd, u0)) + ugen;
    *v = fma(vc, cs, fma(vs, sd, v0)) + vgen;
    const double tt = ddf * t;
    *du = fma(duc, cs, fma(dus, sd, du0)) + tt * (us * cs - uc * sd);
    *dv = tt * (vs * cs - vc * sd);
  };

  // the samples of this wave: the whole series, or (CHUNKED) chunk blockIdx.z of the wide scan's chunking
  int n_lo = 0, n_hi = N;
  if (CHUNKED) {
    const int c = blockIdx.z;
    auto begin = [&](int k) {
      const long n = (P.L0 > 0 && k > 0) ? (long)P.L0 + (long)(k - 1) * P.L : (long)k * P.L;
      return n < N ? (int)n : N;
    };
    n_lo = begin(c);
    n_hi = begin(c + 1);
  }
  double S[COLS], dS[COLS];
#pragma unroll
  for (int c = 0; c < COLS; ++c) { S[c] = 0.0; dS[c] = 0.0; }
  double f = 0.0, df = 0.0, quad = 0.0, dquad = 0.0, dld = 0.0;
  if (CHUNKED && blockIdx.z > 0) {  // the true base state at the chunk's first sample (packed upper triangle | f)
    constexpr int SZP = WMAX * (WMAX + 1) / 2;
    const double* st = P.starts + ((long)blockIdx.y * P.nchunk + blockIdx.z) * (SZP + WMAX);
#pragma unroll
    for (int c = 0; c < COLS; ++c) S[c] = st[sym(row, seg * COLS + c)];
    f = st[SZP + row];
  }
  LogProduct lp;
  lp.init();
  int flag = 0;

  // 64-sample register tiles (t with two samples of look-ahead)
  double tv = n_lo + lane < N ? P.t[n_lo + lane] : 0.0;
  double tv2 = n_lo + lane + 64 < N ? P.t[n_lo + lane + 64] : 0.0;
  double dv_ = n_lo + lane < N ? P.diag[n_lo + lane] : 0.0;
  double yv = n_lo + lane < N ? P.y[n_lo + lane] : 0.0;
  double av = (has_general && n_lo + lane < N) ? P.A[n_lo + lane] : 0.0;
  auto t_at = [&](int k) { return k < 64 ? lane_value(tv, k) : lane_value(tv2, k - 64); };

  // general rows: u, v of sample n are fetched during step n - 2
  double ug1 = 0.0, vg1 = 0.0, ug2 = 0.0, vg2 = 0.0;
  if (ug) {
    ug1 = n_lo + 1 < N ? ug[n_lo + 1] : 0.0; vg1 = n_lo + 1 < N ? vg[n_lo + 1] : 0.0;
    ug2 = n_lo + 2 < N ? ug[n_lo + 2] : 0.0; vg2 = n_lo + 2 < N ? vg[n_lo + 2] : 0.0;
  }
  double u, du, v, dv, phi, dphi;
  features(t_at(0), n_lo + 1 < N ? t_at(1) - t_at(0) : 0.0, (ug && n_lo < N) ? ug[n_lo] : 0.0, (ug && n_lo < N) ? vg[n_lo] : 0.0,
           &u, &du, &v, &dv, &phi, &dphi);
  if (writer) { ubuf[n_lo & 1][row] = u; dubuf[n_lo & 1][row] = du; pbuf[n_lo & 1][row] = phi; dpbuf[n_lo & 1][row] = dphi; }

  for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
    const int nend = (n_hi - n0 < 64) ? n_hi - n0 : 64;
    for (int k = 0; k < nend; ++k) {
      const int n = n0 + k, cur = n & 1;
      const double diag_n = lane_value(dv_, k), y_n = lane_value(yv, k);
      double a_n = ((diag_n + sum_ar) + sum_ac) + jitter;
      if (has_general) a_n += lane_value(av, k);  // cholesky.h:99

      // next sample's features and their tangents: published one step ahead
      double u1 = 0.0, du1 = 0.0, v1 = 0.0, dv1 = 0.0, phi1 = 1.0, dphi1 = 0.0;
      if (n + 1 < N) {
        const double t1 = t_at(k + 1);
        const double dx1 = (n + 2 < N) ? t_at(k + 2) - t1 : 0.0;
        features(t1, dx1, ug1, vg1, &u1, &du1, &v1, &dv1, &phi1, &dphi1);
        if (writer) {
          ubuf[cur ^ 1][row] = u1; dubuf[cur ^ 1][row] = du1;
          pbuf[cur ^ 1][row] = phi1; dpbuf[cur ^ 1][row] = dphi1;
        }
        ug1 = ug2; vg1 = vg2;
        if (ug && n + 3 < N) { ug2 = ug[n + 3]; vg2 = vg[n + 3]; }
      }

      double q = 0.0, dq = 0.0;
      {
        const double2* uv = reinterpret_cast<const double2*>(&ubuf[cur][seg * COLS]);
        const double2* duv = reinterpret_cast<const double2*>(&dubuf[cur][seg * COLS]);
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 uu = uv[c], dd = duv[c];
          q = fma(S[2 * c], uu.x, q);
          q = fma(S[2 * c + 1], uu.y, q);
          dq = fma(dS[2 * c], uu.x, fma(S[2 * c], dd.x, dq));
          dq = fma(dS[2 * c + 1], uu.y, fma(S[2 * c + 1], dd.y, dq));
        }
      }
      if (LPR >= 2) { q = dpp_add<DPP_QUAD_XOR1>(q); dq = dpp_add<DPP_QUAD_XOR1>(dq); }
      if (LPR >= 4) { q = dpp_add<DPP_QUAD_XOR2>(q); dq = dpp_add<DPP_QUAD_XOR2>(dq); }
      double s, ds, ub, dub;
      if constexpr (LPR >= 2) {
        // two sums per butterfly tree (first lane of a row: the q-terms, second lane: the f-terms), completed across the
        // 16-lane rows with the permlane swaps and delivered to every lane as vector values (clr_wide.h: row_sum2_all)
        row_sum2_all<LPR>(u * (seg == 0 ? q : f), seg, &s, &ub);
        row_sum2_all<LPR>(seg == 0 ? fma(du, q, u * dq) : fma(du, f, u * df), seg, &ds, &dub);
      } else {
        s = row_sum_all<LPR>(u * q); ds = row_sum_all<LPR>(fma(du, q, u * dq));
        ub = row_sum_all<LPR>(u * f); dub = row_sum_all<LPR>(fma(du, f, u * df));
      }
      const double D = a_n - s, dD = da - ds;
      const double invD = 1.0 / D;
      const double x = y_n - ub, dx = -dub;
      if (n >= 1 && D < 0.0) flag = 1;  // cholesky.h:176 (sample 0 is never checked)
      lp.mul(D);
      dld = fma(dD, invD, dld);
      const double xs = x * invD;
      quad = fma(x, xs, quad);
      dquad += (2.0 * dx - xs * dD) * xs;

      const double z = v - q, dz = dv - dq;
      const double w = z * invD;
      const double dw = (dz - w * dD) * invD;
      if (writer) { wbuf[row] = phi * w; dwbuf[row] = fma(dphi, w, phi * dw); }
      {
        const double2* pv = reinterpret_cast<const double2*>(&pbuf[cur][seg * COLS]);
        const double2* dpv = reinterpret_cast<const double2*>(&dpbuf[cur][seg * COLS]);
        const double2* wv = reinterpret_cast<const double2*>(&wbuf[seg * COLS]);
        const double2* dwv = reinterpret_cast<const double2*>(&dwbuf[seg * COLS]);
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 pk = pv[c], dpk = dpv[c], pw = wv[c], dpw = dwv[c];
          {
            const double inner = fma(z, pw.x, pk.x * S[2 * c]);
            const double dinner = fma(dpk.x, S[2 * c], fma(pk.x, dS[2 * c], fma(dz, pw.x, z * dpw.x)));
            dS[2 * c] = fma(dphi, inner, phi * dinner);
            S[2 * c] = phi * inner;
          }
          {
            const double inner = fma(z, pw.y, pk.y * S[2 * c + 1]);
            const double dinner =
                fma(dpk.y, S[2 * c + 1], fma(pk.y, dS[2 * c + 1], fma(dz, pw.y, z * dpw.y)));
            dS[2 * c + 1] = fma(dphi, inner, phi * dinner);
            S[2 * c + 1] = phi * inner;
          }
        }
      }
      {
        const double g = fma(w, x, f);
        const double dg = df + fma(dw, x, w * dx);
        df = fma(dphi, g, phi * dg);
        f = phi * g;
      }
      u = u1; du = du1; v = v1; dv = dv1; phi = phi1; dphi = dphi1;
    }
    const int m = n0 + 64 + lane;
    tv = tv2;
    tv2 = m + 64 < N ? P.t[m + 64] : 0.0;
    dv_ = m < N ? P.diag[m] : 0.0;
    yv = m < N ? P.y[m] : 0.0;
    av = (has_general && m < N) ? P.A[m] : 0.0;
  }
  if (CHUNKED) {  // the chunk's record for this direction: dS_end | df_end | d log det | d quad (zero tangent start)
    const int NGc = 1 + 2 * JR + 4 * JC;
    double* o = P.rec + (((long)blockIdx.y * P.nchunk + blockIdx.z) * NGc + blockIdx.x) * ((long)WMAX * WMAX + WMAX + 2);
#pragma unroll
    for (int c = 0; c < COLS; ++c) o[row * WMAX + seg * COLS + c] = dS[c];
    if (writer) o[WMAX * WMAX + row] = df;
    if (lane == 0) { o[WMAX * WMAX + WMAX] = dld; o[WMAX * WMAX + WMAX + 1] = dquad; }
    return;
  }
  if (lane == 0) {
    P.out_grad[blockIdx.x] = -0.5 * (dquad + dld);
    if (blockIdx.x == 0) {
      const double ld = lp.log_value();
      P.out_status[0] = flag ? CLR_NOT_POSITIVE_DEFINITE : CLR_OK;
      P.out_value[0] = -0.5 * (quad + ld + 3.14159265358979323846 * log((double)N));  // solver.cpp:415
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The chunked tangent pass with TWO directions per wave (round 4): a wave of the kernel above spends ~40 % of a step on
// the base recurrence (features, q, the S update, the pivot) that every direction of the chunk repeats.  Here a wave
// carries directions 2 x and 2 x + 1 on ONE base recurrence: NG / 2 waves per (problem, chunk) instead of NG, three
// packed reduction trees per step instead of four, the base's LDS exchanges shared.  Same arithmetic per direction.
// blockIdx = (pair of directions, problem, chunk); record layout as wide_grad_kernel<.., CHUNKED>.
// ---------------------------------------------------------------------------------------------------------------
template <int WMAX, bool FAST>
__global__ void __launch_bounds__(64) wide_grad2_kernel(const GradParams Pin) {
  GradParams P = Pin;
  const long b = blockIdx.y;
  if (Pin.only_level && Pin.only_level[b] >= 2) return;  // (left to the sequential form)
  const int NG = 1 + 2 * Pin.J_real + 4 * Pin.J_comp;
  if (P.A) P.A += b * Pin.A_stride;
  if (P.U) { P.U += b * Pin.U_stride; P.V += b * Pin.V_stride; }
  P.a_real += b * Pin.J_real; P.c_real += b * Pin.J_real;
  P.a_comp += b * Pin.J_comp; P.b_comp += b * Pin.J_comp; P.c_comp += b * Pin.J_comp; P.d_comp += b * Pin.J_comp;
  P.jitter = Pin.jitter_b[b];
  P.t += b * Pin.t_stride; P.diag += b * Pin.diag_stride; P.y += b * Pin.y_stride;
  using G = WideGeom<WMAX>;
  constexpr int LPR = G::LPR, COLS = G::COLS;
  static_assert(LPR >= 2, "the chunked gradient covers the padded widths 16 and 32");
  __shared__ __attribute__((aligned(16))) double ubuf[2][WMAX], pbuf[2][WMAX], wbuf[WMAX];
  __shared__ __attribute__((aligned(16))) double dubuf[2][2][WMAX], dpbuf[2][2][WMAX], dwbuf[2][WMAX];  // [direction]...
  const int lane = threadIdx.x;
  const int row = lane / LPR, seg = lane % LPR;
  const bool writer = seg == 0;
  const int JR = P.J_real, JC = P.J_comp, JG = P.J_general, N = P.N;
  const int Wc = JR + 2 * JC;

  double u0 = 0.0, uc = 0.0, us = 0.0, v0 = 0.0, vc = 0.0, vs = 0.0, cdec = 0.0, dfreq = 0.0;
  int pair = -1;
  bool cosrow = false;
  const double *ug = nullptr, *vg = nullptr;
  if (row < JR) {
    u0 = P.a_real[row]; v0 = 1.0; cdec = P.c_real[row];
  } else if (row < Wc) {
    pair = (row - JR) >> 1;
    cosrow = ((row - JR) & 1) == 0;
    const double a = P.a_comp[pair], bb = P.b_comp[pair];
    if (cosrow) { uc = a; us = bb; vc = 1.0; } else { uc = -bb; us = a; vs = 1.0; }
    cdec = P.c_comp[pair];
    dfreq = P.d_comp[pair];
  } else if (row < Wc + JG) {
    ug = P.U + (long)(row - Wc) * N;
    vg = P.V + (long)(row - Wc) * N;
  }
  // the two directions (solver.cpp:379-406: jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp); a direction past
  // the last one is the zero direction
  double du0[2] = {0.0, 0.0}, duc[2] = {0.0, 0.0}, dus[2] = {0.0, 0.0}, dcf[2] = {0.0, 0.0}, ddf[2] = {0.0, 0.0}, da[2] = {0.0, 0.0};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    int q = 2 * blockIdx.x + d;
    if (q >= NG) continue;
    if (q == 0) {
      da[d] = 1.0;
    } else if ((q -= 1) < JR) {
      da[d] = 1.0;
      if (row == q) du0[d] = 1.0;
    } else if ((q -= JR) < JR) {
      if (row == q) dcf[d] = 1.0;
    } else if ((q -= JR) < JC) {
      da[d] = 1.0;
      if (pair == q) { if (cosrow) duc[d] = 1.0; else dus[d] = 1.0; }
    } else if ((q -= JC) < JC) {
      if (pair == q) { if (cosrow) dus[d] = 1.0; else duc[d] = -1.0; }
    } else if ((q -= JC) < JC) {
      if (pair == q) dcf[d] = 1.0;
    } else {
      q -= JC;
      if (pair == q) ddf[d] = 1.0;
    }
  }
  double sum_ar = 0.0, sum_ac = 0.0;  // cholesky.h:98
  for (int j = 0; j < JR; ++j) sum_ar += P.a_real[j];
  for (int j = 0; j < JC; ++j) sum_ac += P.a_comp[j];
  const double jitter = P.jitter;
  const bool has_general = P.A != nullptr;

  auto features = [&](double t, double dx, double ugen, double vgen, double* u, double* du, double* v, double* dv,
                      double* phi, double* dphi) {
    double sd, cs;
    sincos_phase<FAST>(dfreq * t, &sd, &cs);
    const double x = -cdec * dx;
    const double e = CLR_WAVE_ALL(fabs(x) < 0.0078125) ? exp_small(x) : exp(x);
    *phi = e;
    *u = fma(uc, cs, fma(us, sd, u0)) + ugen;
    *v = fma(vc, cs, fma(vs, sd, v0)) + vgen;
    const double ru = us * cs - uc * sd, rv = vs * cs - vc * sd;  // d u / d(phase), d v / d(phase)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      dphi[d] = -(dcf[d] * dx) * e;
      const double tt = ddf[d] * t;
      du[d] = fma(duc[d], cs, fma(dus[d], sd, du0[d])) + tt * ru;
      dv[d] = tt * rv;
    }
  };

  const int c = blockIdx.z;
  auto begin = [&](int k) {
    const long n = (P.L0 > 0 && k > 0) ? (long)P.L0 + (long)(k - 1) * P.L : (long)k * P.L;
    return n < N ? (int)n : N;
  };
  const int n_lo = begin(c), n_hi = begin(c + 1);
  double S[COLS], dS[2][COLS];
#pragma unroll
  for (int k = 0; k < COLS; ++k) { S[k] = 0.0; dS[0][k] = 0.0; dS[1][k] = 0.0; }
  double f = 0.0, df[2] = {0.0, 0.0}, dquad[2] = {0.0, 0.0}, dld[2] = {0.0, 0.0};
  if (c > 0) {  // the true base state at the chunk's first sample (packed upper triangle | f)
    constexpr int SZP = WMAX * (WMAX + 1) / 2;
    const double* st = P.starts + ((long)b * P.nchunk + c) * (SZP + WMAX);
#pragma unroll
    for (int k = 0; k < COLS; ++k) S[k] = st[sym(row, seg * COLS + k)];
    f = st[SZP + row];
  }

  double tv = n_lo + lane < N ? P.t[n_lo + lane] : 0.0;
  double tv2 = n_lo + lane + 64 < N ? P.t[n_lo + lane + 64] : 0.0;
  double dv_ = n_lo + lane < N ? P.diag[n_lo + lane] : 0.0;
  dv_ = ((dv_ + sum_ar) + sum_ac) + jitter;                       // K(0) of the tile's samples (cholesky.h:98-99)
  if (has_general && n_lo + lane < N) dv_ += P.A[n_lo + lane];
  double yv = n_lo + lane < N ? P.y[n_lo + lane] : 0.0;
  auto t_at = [&](int k) { return k < 64 ? lane_value(tv, k) : lane_value(tv2, k - 64); };

  double ug1 = 0.0, vg1 = 0.0, ug2 = 0.0, vg2 = 0.0;
  if (ug) {
    ug1 = n_lo + 1 < N ? ug[n_lo + 1] : 0.0; vg1 = n_lo + 1 < N ? vg[n_lo + 1] : 0.0;
    ug2 = n_lo + 2 < N ? ug[n_lo + 2] : 0.0; vg2 = n_lo + 2 < N ? vg[n_lo + 2] : 0.0;
  }
  double u, v, phi, du[2], dv[2], dphi[2];
  features(t_at(0), n_lo + 1 < N ? t_at(1) - t_at(0) : 0.0, (ug && n_lo < N) ? ug[n_lo] : 0.0, (ug && n_lo < N) ? vg[n_lo] : 0.0,
           &u, du, &v, dv, &phi, dphi);
  if (writer) {
    ubuf[n_lo & 1][row] = u; pbuf[n_lo & 1][row] = phi;
#pragma unroll
    for (int d = 0; d < 2; ++d) { dubuf[d][n_lo & 1][row] = du[d]; dpbuf[d][n_lo & 1][row] = dphi[d]; }
  }

  for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
    const int nend = (n_hi - n0 < 64) ? n_hi - n0 : 64;
    for (int k = 0; k < nend; ++k) {
      const int n = n0 + k, cur = n & 1;
      const double a_n = lane_value(dv_, k), y_n = lane_value(yv, k);

      double u1 = 0.0, v1 = 0.0, phi1 = 1.0, du1[2] = {0.0, 0.0}, dv1[2] = {0.0, 0.0}, dphi1[2] = {0.0, 0.0};
      if (n + 1 < N) {
        const double t1 = t_at(k + 1);
        const double dx1 = (n + 2 < N) ? t_at(k + 2) - t1 : 0.0;
        features(t1, dx1, ug1, vg1, &u1, du1, &v1, dv1, &phi1, dphi1);
        if (writer) {
          ubuf[cur ^ 1][row] = u1; pbuf[cur ^ 1][row] = phi1;
#pragma unroll
          for (int d = 0; d < 2; ++d) { dubuf[d][cur ^ 1][row] = du1[d]; dpbuf[d][cur ^ 1][row] = dphi1[d]; }
        }
        ug1 = ug2; vg1 = vg2;
        if (ug && n + 3 < N) { ug2 = ug[n + 3]; vg2 = vg[n + 3]; }
      }

      double q = 0.0, dq[2] = {0.0, 0.0};
      {
        const double2* uv = reinterpret_cast<const double2*>(&ubuf[cur][seg * COLS]);
        const double2* duv0 = reinterpret_cast<const double2*>(&dubuf[0][cur][seg * COLS]);
        const double2* duv1 = reinterpret_cast<const double2*>(&dubuf[1][cur][seg * COLS]);
#pragma unroll
        for (int j = 0; j < COLS / 2; ++j) {
          const double2 uu = uv[j], d0 = duv0[j], d1 = duv1[j];
          q = fma(S[2 * j], uu.x, q);
          q = fma(S[2 * j + 1], uu.y, q);
          dq[0] = fma(dS[0][2 * j], uu.x, fma(S[2 * j], d0.x, dq[0]));
          dq[0] = fma(dS[0][2 * j + 1], uu.y, fma(S[2 * j + 1], d0.y, dq[0]));
          dq[1] = fma(dS[1][2 * j], uu.x, fma(S[2 * j], d1.x, dq[1]));
          dq[1] = fma(dS[1][2 * j + 1], uu.y, fma(S[2 * j + 1], d1.y, dq[1]));
        }
      }
      q = dpp_add<DPP_QUAD_XOR1>(q); dq[0] = dpp_add<DPP_QUAD_XOR1>(dq[0]); dq[1] = dpp_add<DPP_QUAD_XOR1>(dq[1]);
      if (LPR >= 4) { q = dpp_add<DPP_QUAD_XOR2>(q); dq[0] = dpp_add<DPP_QUAD_XOR2>(dq[0]); dq[1] = dpp_add<DPP_QUAD_XOR2>(dq[1]); }
      double s, ub, ds[2], dub[2];
      row_sum2_all<LPR>(u * (seg == 0 ? q : f), seg, &s, &ub);
#pragma unroll
      for (int d = 0; d < 2; ++d)
        row_sum2_all<LPR>(seg == 0 ? fma(du[d], q, u * dq[d]) : fma(du[d], f, u * df[d]), seg, &ds[d], &dub[d]);
      const double D = a_n - s;
      const double invD = 1.0 / D;
      const double x = y_n - ub;
      const double xs = x * invD;
      const double z = v - q;
      const double w = z * invD;
      double dz[2], dw[2], dxx[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const double dD = da[d] - ds[d];
        dxx[d] = -dub[d];
        dld[d] = fma(dD, invD, dld[d]);
        dquad[d] += (2.0 * dxx[d] - xs * dD) * xs;
        dz[d] = dv[d] - dq[d];
        dw[d] = (dz[d] - w * dD) * invD;
      }
      if (writer) {
        wbuf[row] = phi * w;
#pragma unroll
        for (int d = 0; d < 2; ++d) dwbuf[d][row] = fma(dphi[d], w, phi * dw[d]);
      }
      {
        const double2* pv = reinterpret_cast<const double2*>(&pbuf[cur][seg * COLS]);
        const double2* wv = reinterpret_cast<const double2*>(&wbuf[seg * COLS]);
        const double2* dpv0 = reinterpret_cast<const double2*>(&dpbuf[0][cur][seg * COLS]);
        const double2* dpv1 = reinterpret_cast<const double2*>(&dpbuf[1][cur][seg * COLS]);
        const double2* dwv0 = reinterpret_cast<const double2*>(&dwbuf[0][seg * COLS]);
        const double2* dwv1 = reinterpret_cast<const double2*>(&dwbuf[1][seg * COLS]);
#pragma unroll
        for (int j = 0; j < COLS / 2; ++j) {
          const double2 pk = pv[j], pw = wv[j], dpk0 = dpv0[j], dpk1 = dpv1[j], dpw0 = dwv0[j], dpw1 = dwv1[j];
          {
            const double inner = fma(z, pw.x, pk.x * S[2 * j]);
            const double di0 = fma(dpk0.x, S[2 * j], fma(pk.x, dS[0][2 * j], fma(dz[0], pw.x, z * dpw0.x)));
            const double di1 = fma(dpk1.x, S[2 * j], fma(pk.x, dS[1][2 * j], fma(dz[1], pw.x, z * dpw1.x)));
            dS[0][2 * j] = fma(dphi[0], inner, phi * di0);
            dS[1][2 * j] = fma(dphi[1], inner, phi * di1);
            S[2 * j] = phi * inner;
          }
          {
            const double inner = fma(z, pw.y, pk.y * S[2 * j + 1]);
            const double di0 = fma(dpk0.y, S[2 * j + 1], fma(pk.y, dS[0][2 * j + 1], fma(dz[0], pw.y, z * dpw0.y)));
            const double di1 = fma(dpk1.y, S[2 * j + 1], fma(pk.y, dS[1][2 * j + 1], fma(dz[1], pw.y, z * dpw1.y)));
            dS[0][2 * j + 1] = fma(dphi[0], inner, phi * di0);
            dS[1][2 * j + 1] = fma(dphi[1], inner, phi * di1);
            S[2 * j + 1] = phi * inner;
          }
        }
      }
      {
        const double g = fma(w, x, f);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const double dg = df[d] + fma(dw[d], x, w * dxx[d]);
          df[d] = fma(dphi[d], g, phi * dg);
        }
        f = phi * g;
      }
      u = u1; v = v1; phi = phi1;
#pragma unroll
      for (int d = 0; d < 2; ++d) { du[d] = du1[d]; dv[d] = dv1[d]; dphi[d] = dphi1[d]; }
    }
    const int m = n0 + 64 + lane;
    tv = tv2;
    tv2 = m + 64 < N ? P.t[m + 64] : 0.0;
    dv_ = m < N ? P.diag[m] : 0.0;
    dv_ = ((dv_ + sum_ar) + sum_ac) + jitter;
    if (has_general && m < N) dv_ += P.A[m];
    yv = m < N ? P.y[m] : 0.0;
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int q = 2 * blockIdx.x + d;
    if (q >= NG) continue;
    double* o = P.rec + (((long)b * P.nchunk + c) * NG + q) * ((long)WMAX * WMAX + WMAX + 2);
#pragma unroll
    for (int k = 0; k < COLS; ++k) o[row * WMAX + seg * COLS + k] = dS[d][k];
    if (writer) o[WMAX * WMAX + row] = df[d];
    if (lane == 0) { o[WMAX * WMAX + WMAX] = dld[d]; o[WMAX * WMAX + WMAX + 1] = dquad[d]; }
  }
}

}  // namespace

void launch_grad_chunked(const GradParams& P, hipStream_t s) {
#ifndef CLR_GRAD_TWO_DIRECTIONS
#define CLR_GRAD_TWO_DIRECTIONS 1
#endif
  const int NG = 1 + 2 * P.J_real + 4 * P.J_comp;
  if (P.JP == 64) {  // widths 33..64 (round 6): one direction per wave (a row of S and of dS per lane: 256 registers)
    const dim3 grid(NG, P.B, P.nchunk);
    if (P.fast_trig) hipLaunchKernelGGL((wide_grad_kernel<64, true, true>), grid, dim3(64), 0, s, P);
    else hipLaunchKernelGGL((wide_grad_kernel<64, false, true>), grid, dim3(64), 0, s, P);
    return;
  }
  if (CLR_GRAD_TWO_DIRECTIONS) {  // two directions per wave on one base recurrence
    const dim3 grid2((NG + 1) / 2, P.B, P.nchunk);
    if (P.JP == 16) {
      if (P.fast_trig) hipLaunchKernelGGL((wide_grad2_kernel<16, true>), grid2, dim3(64), 0, s, P);
      else hipLaunchKernelGGL((wide_grad2_kernel<16, false>), grid2, dim3(64), 0, s, P);
    } else {
      if (P.fast_trig) hipLaunchKernelGGL((wide_grad2_kernel<32, true>), grid2, dim3(64), 0, s, P);
      else hipLaunchKernelGGL((wide_grad2_kernel<32, false>), grid2, dim3(64), 0, s, P);
    }
    return;
  }
  const dim3 grid(NG, P.B, P.nchunk);
  if (P.JP == 16) {
    if (P.fast_trig) hipLaunchKernelGGL((wide_grad_kernel<16, true, true>), grid, dim3(64), 0, s, P);
    else hipLaunchKernelGGL((wide_grad_kernel<16, false, true>), grid, dim3(64), 0, s, P);
  } else {
    if (P.fast_trig) hipLaunchKernelGGL((wide_grad_kernel<32, true, true>), grid, dim3(64), 0, s, P);
    else hipLaunchKernelGGL((wide_grad_kernel<32, false, true>), grid, dim3(64), 0, s, P);
  }
}

void launch_grad(const GradParams& P, hipStream_t s) {
  const int W = P.J_real + 2 * P.J_comp + P.J_general;
  const dim3 grid(1 + 2 * P.J_real + 4 * P.J_comp, P.B > 0 ? P.B : 1);
#define CLR_GO(WM)                                                                           \
  do {                                                                                       \
    if (P.fast_trig) hipLaunchKernelGGL((wide_grad_kernel<WM, true>), grid, dim3(64), 0, s, P); \
    else hipLaunchKernelGGL((wide_grad_kernel<WM, false>), grid, dim3(64), 0, s, P);            \
  } while (0)
  if (W <= 16) CLR_GO(16);
  else if (W <= 32) CLR_GO(32);
  else CLR_GO(64);
#undef CLR_GO
}

}  // namespace clr
