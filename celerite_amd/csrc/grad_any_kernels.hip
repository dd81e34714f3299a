// celerite_amd/csrc/grad_any_kernels.hip -- CholeskySolver.grad_log_likelihood at ANY width (65 .. CLR_MAX_WIDTH_ANY).
//
// The reference differentiates by running its solver template on forward-mode dual numbers (celerite/solver.cpp:347-463)
// and its dynamic-width arm takes any J (cholesky.h:203).  The wave-per-direction kernel of grad_kernels.hip keeps a row
// of S and of dS per lane and stops at width 64.  Above it: ONE WORKGROUP of 1024 threads PER DIRECTION (grid.x = the
// directions of a round), the same tangent recurrence
//     q = S u                      dq = dS u + S du
//     D = a - u.q                  dD = da - (du.q + u.dq)
//     z = v - q ; w = z / D        dz = dv - dq ; dw = (dz - w dD) / D
//     x = y - u.f                  dx = -(du.f + u.df)
//     S <- Phi (S + z w^T) Phi     dS <- the product rule
//     f <- Phi (f + w x)           df <- dPhi (f + w x) + Phi (df + dw x + w dx)
// with S and dS (J x J each, full storage) in a workspace in HBM / L2 and every vector of length J in LDS.  A step is ONE
// pass over the two matrices: wave w owns rows w, w + 16, ...; its lanes stride over the columns, update S_ik and dS_ik
// in place and at once accumulate the NEXT sample's q_i = sum_k S_ik u_k and dq_i (the features of sample n + 1 do not
// depend on the state: they are formed before the pass), so a matrix entry is read once and written once per step.
// Two workgroup barriers per step (the vectors of step n / the pass).  A direction only touches the rows of its own term:
// du, dv, dphi are kept for those (at most two) rows and are zero elsewhere.  General terms (A, U, V) are constants of
// the differentiation: extra rows with u = U[j][n], v = V[j][n], phi = 1.  Completeness, not speed: at width 128 a step
// moves 0.5 MB per direction through L2 / MALL -- the reference's own dual-number recurrence costs 5.5 J^2 (2 J + 1) flops
// per step on one core.
#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_wide.h"

#include <algorithm>

namespace clr {

namespace {

__device__ __forceinline__ double ga_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

constexpr int GA_THREADS = 1024, GA_WAVES = GA_THREADS / 64;

template <bool FAST>
__global__ void __launch_bounds__(GA_THREADS) grad_any_kernel(const GradParams P, double* workspace, int dir0) {
  extern __shared__ __attribute__((aligned(16))) double ga_lds[];
  const int JR = P.J_real, JC = P.J_comp, JG = P.J_general, N = P.N;
  const int Wc = JR + 2 * JC, J = Wc + JG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = dir0 + blockIdx.x;
  // LDS: u, v, phi of samples n / n + 1 (two buffers), q, dq, f, df (two buffers each), pw = phi w, dpw
  double* fu = ga_lds;            // [2][J]
  double* fv = fu + 2 * J;        // [2][J]
  double* fp = fv + 2 * J;        // [2][J]
  double* q = fp + 2 * J;         // [2][J]
  double* dq = q + 2 * J;         // [2][J]
  double* f = dq + 2 * J;         // [2][J]
  double* df = f + 2 * J;         // [2][J]
  double* pw = df + 2 * J;        // [J]
  double* dpw = pw + J;           // [J]
  __shared__ double dfe[2][3][2];  // [buffer][du, dv, dphi][row r0, r0 + 1]
  double* S = workspace + (size_t)blockIdx.x * 2 * J * J;
  double* dS = S + (size_t)J * J;

  // ---- this workgroup's direction (solver.cpp:379-406: jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp) ----
  int r0 = -1, nrow = 0, kind = 0;  // kind: 0 jitter, 1 a_real, 2 c_real, 3 a_comp, 4 b_comp, 5 c_comp, 6 d_comp
  double da = 0.0;
  {
    int g = dir;
    if (g == 0) { da = 1.0; }
    else if ((g -= 1) < JR) { kind = 1; da = 1.0; r0 = g; nrow = 1; }
    else if ((g -= JR) < JR) { kind = 2; r0 = g; nrow = 1; }
    else if ((g -= JR) < JC) { kind = 3; da = 1.0; r0 = JR + 2 * g; nrow = 2; }
    else if ((g -= JC) < JC) { kind = 4; r0 = JR + 2 * g; nrow = 2; }
    else if ((g -= JC) < JC) { kind = 5; r0 = JR + 2 * g; nrow = 2; }
    else { g -= JC; kind = 6; r0 = JR + 2 * g; nrow = 2; }
  }
  // value of a sparse tangent feature at row k (zero outside the direction's term)
  auto sp = [&](int buf, int what, int k) -> double {
    const int o = k - r0;
    return (o >= 0 && o < nrow) ? dfe[buf][what][o] : 0.0;
  };

  // ---- row constants of thread tid (rows 0 .. J-1) ------------------------------------------------------------------
  const int row = tid;
  double u0 = 0.0, uc = 0.0, us = 0.0, v0 = 0.0, vc = 0.0, vs = 0.0, cdec = 0.0, dfreq = 0.0;
  bool cosrow = false;
  const double *ug = nullptr, *vg = nullptr;
  if (row < JR) {
    u0 = P.a_real[row]; v0 = 1.0; cdec = P.c_real[row];
  } else if (row < Wc) {
    const int pair = (row - JR) >> 1;
    cosrow = ((row - JR) & 1) == 0;
    const double a = P.a_comp[pair], b = P.b_comp[pair];
    if (cosrow) { uc = a; us = b; vc = 1.0; }   // cholesky.h:143,145
    else        { uc = -b; us = a; vs = 1.0; }  // cholesky.h:144,146
    cdec = P.c_comp[pair];
    dfreq = P.d_comp[pair];
  } else if (row < J) {
    ug = P.U + (long)(row - Wc) * N;
    vg = P.V + (long)(row - Wc) * N;
  }
  const bool mine = row >= r0 && row < r0 + nrow;
  double du0 = 0.0, duc = 0.0, dus = 0.0, dcf = 0.0, ddf = 0.0;
  if (mine) {
    if (kind == 1) du0 = 1.0;
    else if (kind == 2 || kind == 5) dcf = 1.0;
    else if (kind == 3) { if (cosrow) duc = 1.0; else dus = 1.0; }
    else if (kind == 4) { if (cosrow) dus = 1.0; else duc = -1.0; }
    else if (kind == 6) ddf = 1.0;
  }
  double sum_ar = 0.0, sum_ac = 0.0;  // cholesky.h:98
  for (int j = 0; j < JR; ++j) sum_ar += P.a_real[j];
  for (int j = 0; j < JC; ++j) sum_ac += P.a_comp[j];
  const bool has_general = P.A != nullptr;

  // features of sample n (and their tangents in this direction) -> buffer b
  auto publish = [&](int n, int b) {
    if (row >= J) return;
    const double t = P.t[n];
    const double dx = n + 1 < N ? P.t[n + 1] - t : 0.0;
    double u, v, phi, du = 0.0, dv = 0.0, dphi = 0.0;
    if (row < Wc) {
      double sd = 0.0, cs = 1.0;
      if (row >= JR) sincos_phase<FAST>(dfreq * t, &sd, &cs);
      const double e = exp(-cdec * dx);
      phi = e;
      u = fma(uc, cs, fma(us, sd, u0));
      v = fma(vc, cs, fma(vs, sd, v0));
      if (mine) {
        dphi = -(dcf * dx) * e;
        const double tt = ddf * t;
        du = fma(duc, cs, fma(dus, sd, du0)) + tt * (us * cs - uc * sd);
        dv = tt * (vs * cs - vc * sd);
      }
    } else {
      u = ug[n]; v = vg[n]; phi = 1.0;
    }
    fu[b * J + row] = u; fv[b * J + row] = v; fp[b * J + row] = phi;
    if (mine) { dfe[b][0][row - r0] = du; dfe[b][1][row - r0] = dv; dfe[b][2][row - r0] = dphi; }
  };

  // ---- start: S = dS = 0, q = dq = f = df = 0 -----------------------------------------------------------------------
  for (long idx = tid; idx < (long)J * J; idx += GA_THREADS) { S[idx] = 0.0; dS[idx] = 0.0; }
  if (row < J) { q[row] = 0.0; dq[row] = 0.0; f[row] = 0.0; df[row] = 0.0; }
  publish(0, 0);
  __syncthreads();

  double logdet = 0.0, dld = 0.0, quad = 0.0, dquad = 0.0;
  int flag = 0;
  for (int n = 0; n < N; ++n) {
    const int cur = n & 1, nxt = cur ^ 1;
    // ---- the vectors of step n ----
    if (n + 1 < N) publish(n + 1, nxt);
    const double* u = fu + cur * J;
    const double* v = fv + cur * J;
    const double* ph = fp + cur * J;
    const double* qc = q + cur * J;
    const double* dqc = dq + cur * J;
    const double* fc = f + cur * J;
    const double* dfc = df + cur * J;
    double s = 0.0, ds = 0.0, ub = 0.0, dub = 0.0;  // u.q, du.q + u.dq, u.f, du.f + u.df (every wave: the same sums)
    for (int k = lane; k < J; k += 64) {
      const double uk = u[k], duk = sp(cur, 0, k);
      s = fma(uk, qc[k], s);
      ds = fma(duk, qc[k], fma(uk, dqc[k], ds));
      ub = fma(uk, fc[k], ub);
      dub = fma(duk, fc[k], fma(uk, dfc[k], dub));
    }
    s = ga_sum(s); ds = ga_sum(ds); ub = ga_sum(ub); dub = ga_sum(dub);
    double a_n = ((P.diag[n] + sum_ar) + sum_ac) + P.jitter;
    if (has_general) a_n += P.A[n];  // cholesky.h:99
    const double D = a_n - s, dD = da - ds;
    const double invD = 1.0 / D;
    const double x = P.y[n] - ub, dx = -dub;
    if (n >= 1 && D < 0.0) flag = 1;  // cholesky.h:176 (sample 0 is never checked)
    logdet += log(D);
    dld = fma(dD, invD, dld);
    const double xs = x * invD;
    quad = fma(x, xs, quad);
    dquad += (2.0 * dx - xs * dD) * xs;
    if (n + 1 == N) break;
    if (row < J) {
      const double z = v[row] - qc[row], dz = sp(cur, 1, row) - dqc[row];
      const double w = z * invD;
      const double dw = (dz - w * dD) * invD;
      const double phi = ph[row], dphi = sp(cur, 2, row);
      pw[row] = phi * w;
      dpw[row] = fma(dphi, w, phi * dw);
      const double g = fma(w, x, fc[row]);
      const double dg = dfc[row] + fma(dw, x, w * dx);
      f[nxt * J + row] = phi * g;
      df[nxt * J + row] = fma(dphi, g, phi * dg);
    }
    __syncthreads();
    // ---- the pass: S, dS updated in place; q, dq of sample n + 1 ----
    const double* un = fu + nxt * J;
    for (int i = wave; i < J; i += GA_WAVES) {
      const double phi_i = ph[i], dphi_i = sp(cur, 2, i);
      const double z_i = v[i] - qc[i], dz_i = sp(cur, 1, i) - dqc[i];
      double aq = 0.0, adq = 0.0;
      double* Si = S + (size_t)i * J;
      double* dSi = dS + (size_t)i * J;
#pragma unroll 4
      for (int k = lane; k < J; k += 64) {
        const double sk = Si[k], dsk = dSi[k];
        const double pk = ph[k], dpk = sp(cur, 2, k), pwk = pw[k], dpwk = dpw[k];
        const double inner = fma(z_i, pwk, pk * sk);
        const double dinner = fma(dpk, sk, fma(pk, dsk, fma(dz_i, pwk, z_i * dpwk)));
        const double sn = phi_i * inner;
        const double dsn = fma(dphi_i, inner, phi_i * dinner);
        Si[k] = sn;
        dSi[k] = dsn;
        const double unk = un[k], dunk = sp(nxt, 0, k);
        aq = fma(sn, unk, aq);
        adq = fma(dsn, unk, fma(sn, dunk, adq));
      }
      aq = ga_sum(aq); adq = ga_sum(adq);
      if (lane == 0) { q[nxt * J + i] = aq; dq[nxt * J + i] = adq; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    P.out_grad[dir] = -0.5 * (dquad + dld);
    if (dir == 0) {
      P.out_status[0] = flag ? CLR_NOT_POSITIVE_DEFINITE : CLR_OK;
      P.out_value[0] = -0.5 * (quad + logdet + 3.14159265358979323846 * log((double)N));  // solver.cpp:415
    }
  }
}

}  // namespace

// directions per launch: one workgroup per CU
int grad_any_round() { return 256; }
size_t grad_any_workspace_doubles(int J, int NG) { return (size_t)std::min(NG, grad_any_round()) * 2 * (size_t)J * J; }

// B == 0 (one problem); any total width up to CLR_MAX_WIDTH_ANY.  Non-zero: the kernel could not be configured (LDS).
int launch_grad_any(const GradParams& P, double* workspace, hipStream_t s) {
  const int J = P.J_real + 2 * P.J_comp + P.J_general, NG = 1 + 2 * P.J_real + 4 * P.J_comp;
  const int lds = 16 * J * (int)sizeof(double);
  const void* fn = P.fast_trig ? reinterpret_cast<const void*>(&grad_any_kernel<true>) : reinterpret_cast<const void*>(&grad_any_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
  for (int d0 = 0; d0 < NG; d0 += grad_any_round()) {
    const int nd = std::min(grad_any_round(), NG - d0);
    if (P.fast_trig) hipLaunchKernelGGL((grad_any_kernel<true>), dim3(nd), dim3(GA_THREADS), lds, s, P, workspace, d0);
    else hipLaunchKernelGGL((grad_any_kernel<false>), dim3(nd), dim3(GA_THREADS), lds, s, P, workspace, d0);
  }
  return 0;
}

}  // namespace clr
