// celerite_amd/csrc/generic_kernels.hip -- see clr_generic_kernels.h.
#include "clr_generic_kernels.h"
#include "clr_wide.h"

#include <math.h>

namespace clr {
namespace {

// sum over the 64 lanes, wave-uniform result: DPP butterflies within the 16-lane rows and four
// v_readlane pairs across them (clr_wide.h) -- a __shfl_xor butterfly is twelve ds_bpermute_b32
// through the LDS crossbar, ~0.3 us per step for the lone wave of these sequential sweeps
__device__ __forceinline__ double wave_sum(double v) { return clr::row_sum<1>(v); }

// Row j of (phi, u~, v~) for the move t_prev -> t (cholesky.h:127-152).
__device__ __forceinline__ void row_features(const GenericProblem& g, int j, int n, double t,
                                             double dx, double& phi, double& u, double& v) {
  if (j < g.J_real) {
    phi = exp(-g.c_real[j] * dx);
    u = g.a_real[j];
    v = 1.0;
  } else if (j < g.J_real + 2 * g.J_comp) {
    const int jj = (j - g.J_real) >> 1;
    const bool odd = (j - g.J_real) & 1;
    const double a = g.a_comp[jj], b = g.b_comp[jj];
    double sd, cd;
    sincos(g.d_comp[jj] * t, &sd, &cd);
    phi = exp(-g.c_comp[jj] * dx);
    u = odd ? (a * sd - b * cd) : (a * cd + b * sd);
    v = odd ? sd : cd;
  } else {
    const int jg = j - g.J_real - 2 * g.J_comp;
    phi = 1.0;
    u = g.U[(long)jg * g.N + n];
    v = g.V[(long)jg * g.N + n];
  }
}

// ---------------------------------------------------------------------------
// Factorisation, any width.  One workgroup; S lives in LDS (upper triangle
// used, column-major S[k + J j], k <= j).  Follows the reference's step order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) factor_generic_kernel(GenericProblem g, double* phi,
                                                             double* u, double* W, double* D,
                                                             int* status, double* log_det) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int J = g.J, N = g.N, tid = threadIdx.x, nt = blockDim.x;
  double* S = reinterpret_cast<double*>(smem);
  double* sphi = S + (long)J * J;
  double* su = sphi + J;
  double* sv = su + J;
  double* swp = sv + J;
  double* sq = swp + J;
  double* sp = sq + J;
  double* sscal = sp + J;  // [0] = D_n, [1] = failure flag

  for (int i = tid; i < J * J; i += nt) S[i] = 0.0;
  if (tid == 0) sscal[1] = 0.0;

  // sample 0: cholesky.h:100-117
  double Dprev = D[0];
  double ld = log(Dprev);
  {
    const double value = 1.0 / Dprev;
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      row_features(g, j, 0, g.t[0], 0.0, ph, uu, vv);
      const double w = vv * value;
      W[j] = w;
      swp[j] = w;
    }
  }
  __syncthreads();

  for (int n = 1; n < N; ++n) {
    const double t = g.t[n], dx = t - g.t[n - 1];
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      row_features(g, j, n, t, dx, ph, uu, vv);
      sphi[j] = ph;
      su[j] = uu;
      sv[j] = vv;
      phi[(long)J * (n - 1) + j] = ph;
      u[(long)J * (n - 1) + j] = uu;
    }
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += nt) {  // cholesky.h:154-160
      const int k = idx % J, j = idx / J;
      if (k <= j) {
        const double xj = Dprev * swp[j];
        S[idx] = sphi[j] * (sphi[k] * (S[idx] + xj * swp[k]));
      }
    }
    __syncthreads();
    for (int j = tid; j < J; j += nt) {  // q = S u~ ; cholesky.h:163-175
      double acc = 0.0;
      for (int k = 0; k < J; ++k) acc += (k <= j ? S[k + (long)J * j] : S[j + (long)J * k]) * su[k];
      sq[j] = acc;
      sp[j] = su[j] * acc;
    }
    __syncthreads();
    if (tid < 64) {
      double part = 0.0;
      for (int j = tid; j < J; j += 64) part += sp[j];
      part = wave_sum(part);
      if (tid == 0) {
        const double Dn = D[n] - part;
        if (Dn < 0.0) sscal[1] = 1.0;  // cholesky.h:176
        sscal[0] = Dn;
        D[n] = Dn;
        ld += log(Dn);
      }
    }
    __syncthreads();
    if (sscal[1] != 0.0) {
      if (tid == 0) { status[0] = 1; log_det[0] = NAN; }
      return;
    }
    const double Dn = sscal[0];
    for (int j = tid; j < J; j += nt) {  // cholesky.h:170-178
      const double w = (sv[j] - sq[j]) / Dn;
      W[(long)J * n + j] = w;
      swp[j] = w;
    }
    Dprev = Dn;
    __syncthreads();
  }
  if (tid == 0) { status[0] = 0; log_det[0] = ld; }
}

// ---------------------------------------------------------------------------
// Batched fused log-likelihood with general terms: factor_generic_kernel's recurrence with the forward sweep of
// dot_solve (cholesky.h:343-357) carried along -- f <- phi (f + W_{n-1} x_{n-1}), x_n = y_n - u~_n . f -- one
// workgroup per problem (blockIdx.x), S in LDS.  status 2 = a pivot D_n < 0 with n >= 1 (cholesky.h:176).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) generic_loglike_batch_kernel(const GenericBatch G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, N = G.N, tid = threadIdx.x, nt = blockDim.x;
  GenericProblem g;
  g.N = N;
  g.J_real = G.J_real; g.J_comp = G.J_comp; g.J_general = G.J_general;
  g.J = G.J_real + 2 * G.J_comp + G.J_general;
  g.a_real = G.a_real + (long)b * G.J_real; g.c_real = G.c_real + (long)b * G.J_real;
  g.a_comp = G.a_comp + (long)b * G.J_comp; g.b_comp = G.b_comp + (long)b * G.J_comp;
  g.c_comp = G.c_comp + (long)b * G.J_comp; g.d_comp = G.d_comp + (long)b * G.J_comp;
  g.U = G.U ? G.U + (long)b * G.U_stride : nullptr;
  g.V = G.V ? G.V + (long)b * G.V_stride : nullptr;
  g.t = G.t + (long)b * G.t_stride;
  const double* diag = G.diag + (long)b * G.diag_stride;
  const double* y = G.y + (long)b * G.y_stride;
  const double* A = G.A ? G.A + (long)b * G.A_stride : nullptr;
  const int J = g.J;
  double* S = reinterpret_cast<double*>(smem);
  double* sphi = S + (long)J * J;
  double* su = sphi + J;
  double* sv = su + J;
  double* swp = sv + J;
  double* sq = swp + J;
  double* sp = sq + J;
  double* sf = sp + J;   // f of dot_solve
  double* sg = sf + J;   // u~ . f partial products
  double* sscal = sg + J;  // [0] = D_n, [1] = failure flag, [2] = x_n

  // K(0) summed in the reference's order (cholesky.h:98-99): diag + sum a_real + sum a_comp + jitter (+ A)
  double sum_ar = 0.0, sum_ac = 0.0;
  for (int j = 0; j < G.J_real; ++j) sum_ar += g.a_real[j];
  for (int j = 0; j < G.J_comp; ++j) sum_ac += g.a_comp[j];
  const double jit = G.jitter[b];
  auto diagonal = [&](int n) {
    double d = ((diag[n] + sum_ar) + sum_ac) + jit;
    if (A) d += A[n];
    return d;
  };

  for (int i = tid; i < J * J; i += nt) S[i] = 0.0;
  for (int j = tid; j < J; j += nt) sf[j] = 0.0;
  if (tid == 0) sscal[1] = 0.0;

  // sample 0: cholesky.h:100-117, :346-347
  double Dprev = diagonal(0);
  double ld = log(Dprev);
  double xm1 = y[0];
  double quad = xm1 * (xm1 / Dprev);
  {
    const double value = 1.0 / Dprev;
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      row_features(g, j, 0, g.t[0], 0.0, ph, uu, vv);
      swp[j] = vv * value;
    }
  }
  __syncthreads();

  for (int n = 1; n < N; ++n) {
    const double t = g.t[n], dx = t - g.t[n - 1];
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      row_features(g, j, n, t, dx, ph, uu, vv);
      sphi[j] = ph;
      su[j] = uu;
      sv[j] = vv;
      const double f = ph * (sf[j] + swp[j] * xm1);  // cholesky.h:350-352
      sf[j] = f;
      sg[j] = uu * f;
    }
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += nt) {  // cholesky.h:154-160
      const int k = idx % J, j = idx / J;
      if (k <= j) {
        const double xj = Dprev * swp[j];
        S[idx] = sphi[j] * (sphi[k] * (S[idx] + xj * swp[k]));
      }
    }
    __syncthreads();
    for (int j = tid; j < J; j += nt) {  // q = S u~ ; cholesky.h:163-175
      double acc = 0.0;
      for (int k = 0; k < J; ++k) acc += (k <= j ? S[k + (long)J * j] : S[j + (long)J * k]) * su[k];
      sq[j] = acc;
      sp[j] = su[j] * acc;
    }
    __syncthreads();
    if (tid < 64) {
      double part = 0.0, xpart = 0.0;
      for (int j = tid; j < J; j += 64) { part += sp[j]; xpart += sg[j]; }
      part = wave_sum(part);
      xpart = wave_sum(xpart);
      if (tid == 0) {
        const double Dn = diagonal(n) - part;
        if (Dn < 0.0) sscal[1] = 1.0;  // cholesky.h:176
        sscal[0] = Dn;
        sscal[2] = y[n] - xpart;       // :353-354
      }
    }
    __syncthreads();
    if (sscal[1] != 0.0) {
      if (tid == 0) {
        G.out_status[b] = 2;
        G.out_ll[b] = -INFINITY;
        G.out_logdet[b] = NAN;
        G.out_quad[b] = NAN;
      }
      return;
    }
    const double Dn = sscal[0], x = sscal[2];
    ld += log(Dn);
    quad += x * x / Dn;  // :356
    xm1 = x;
    for (int j = tid; j < J; j += nt) swp[j] = (sv[j] - sq[j]) / Dn;  // cholesky.h:170-178
    Dprev = Dn;
    __syncthreads();
  }
  if (tid == 0) {
    G.out_status[b] = 0;
    G.out_logdet[b] = ld;
    G.out_quad[b] = quad;
    double ll = -0.5 * (quad + ld + N * 1.8378770664093453);
    if (!isfinite(ld) || !isfinite(ll)) ll = -INFINITY;  // celerite.py:211-218
    G.out_ll[b] = ll;
  }
}

// J == 0: cholesky.h:90-95.
__global__ void __launch_bounds__(256) diag_only_kernel(int N, const double* diag, double jitter,
                                                        double* D, double* log_det) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double d = diag[n] + jitter;
    D[n] = d;
    acc += log(d);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) log_det[0] = red[0];
}

// ---------------------------------------------------------------------------
// Sweeps over a stored factor: one wave per right-hand side, rows j0 = lane and
// j1 = lane + 64 in registers.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) dot_solve_kernel(int N, int J, const double* phi,
                                                       const double* u, const double* W,
                                                       const double* D, const double* b,
                                                       double* out) {
  // rows j0 = lane and j1 = lane + 64; the factor of KB steps is held in registers and the NEXT KB
  // steps are in flight while they are consumed: one step of look-ahead left the ~0.3 us L2 round trip
  // on every step's critical path (30 ms at N = 1e5 whatever the width)
  constexpr int KB = 8;
  const int lane = threadIdx.x, j0 = lane, j1 = lane + 64;
  const bool h0 = j0 < J, h1 = j1 < J;
  double f0 = 0.0, f1 = 0.0;
  double xm1 = b[0];
  double result = xm1 * (xm1 / D[0]);  // cholesky.h:347
  double p0[KB], u0[KB], w0[KB], p1[KB], u1[KB], w1[KB], bt = 0.0, dt = 1.0;
  auto fetch = [&](int n0) {  // steps n0 .. n0 + KB - 1 read row n - 1 of phi, u, W and b[n], D[n]
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int n = n0 + k;
      const long base = (long)J * (n - 1);
      const bool ok = n < N;
      p0[k] = (ok && h0) ? phi[base + j0] : 0.0; u0[k] = (ok && h0) ? u[base + j0] : 0.0; w0[k] = (ok && h0) ? W[base + j0] : 0.0;
      p1[k] = (ok && h1) ? phi[base + j1] : 0.0; u1[k] = (ok && h1) ? u[base + j1] : 0.0; w1[k] = (ok && h1) ? W[base + j1] : 0.0;
    }
    bt = (lane < KB && n0 + lane < N) ? b[n0 + lane] : 0.0;
    dt = (lane < KB && n0 + lane < N) ? D[n0 + lane] : 1.0;
  };
  fetch(1);
  for (int n0 = 1; n0 < N; n0 += KB) {
    double cp0[KB], cu0[KB], cw0[KB], cp1[KB], cu1[KB], cw1[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { cp0[k] = p0[k]; cu0[k] = u0[k]; cw0[k] = w0[k]; cp1[k] = p1[k]; cu1[k] = u1[k]; cw1[k] = w1[k]; }
    const double cb = bt, cd = dt;
    if (n0 + KB < N) fetch(n0 + KB);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (n0 + k < N) {
        double part = 0.0;  // cholesky.h:350-354
        if (h0) { f0 = cp0[k] * (f0 + cw0[k] * xm1); part += cu0[k] * f0; }
        if (h1) { f1 = cp1[k] * (f1 + cw1[k] * xm1); part += cu1[k] * f1; }
        const double x = clr::lane_value(cb, k) - wave_sum(part);
        xm1 = x;
        result += x * x / clr::lane_value(cd, k);  // :356
      }
    }
  }
  if (threadIdx.x == 0) out[0] = result;
}

// KB steps of the factor (rows j0 = lane, j1 = lane + 64 of phi, u, W at factor rows r0, r0 + dir, ...) held in
// registers while the next KB steps are in flight, as in dot_solve_kernel.
template <int KB>
struct FactorRows {
  double p0[KB], u0[KB], w0[KB], p1[KB], u1[KB], w1[KB];
  __device__ __forceinline__ void fetch(const double* phi, const double* u, const double* W, int J, long r0, int dir,
                                        long rmin, long rmax, int j0, int j1, bool h0, bool h1) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const long r = r0 + (long)dir * k;
      const bool ok = r >= rmin && r <= rmax;
      const long base = (long)J * r;
      p0[k] = (ok && h0) ? phi[base + j0] : 0.0; u0[k] = (ok && h0) ? u[base + j0] : 0.0; w0[k] = (ok && h0) ? W[base + j0] : 0.0;
      p1[k] = (ok && h1) ? phi[base + j1] : 0.0; u1[k] = (ok && h1) ? u[base + j1] : 0.0; w1[k] = (ok && h1) ? W[base + j1] : 0.0;
    }
  }
};

__global__ void __launch_bounds__(64) solve_kernel(int N, int J, const double* phi,
                                                   const double* u, const double* W,
                                                   const double* D, const double* b, double* x) {
  constexpr int KB = 8;
  const int lane = threadIdx.x, j0 = lane, j1 = lane + 64;
  const bool h0 = j0 < J, h1 = j1 < J;
  const double* bk = b + (long)blockIdx.x * N;
  double* xk = x + (long)blockIdx.x * N;

  // forward, cholesky.h:240-248 (x holds the undivided values for now): step n reads factor row n - 1
  double f0 = 0.0, f1 = 0.0;
  double xm1 = bk[0];
  if (lane == 0) xk[0] = xm1;
  {
    FactorRows<KB> nx, cur;
    double bt = 0.0;
    nx.fetch(phi, u, W, J, 0, +1, 0, (long)N - 2, j0, j1, h0, h1);
    bt = (lane < KB && 1 + lane < N) ? bk[1 + lane] : 0.0;
    for (int n0 = 1; n0 < N; n0 += KB) {
      cur = nx;
      const double cb = bt;
      if (n0 + KB < N) {
        nx.fetch(phi, u, W, J, (long)n0 + KB - 1, +1, 0, (long)N - 2, j0, j1, h0, h1);
        bt = (lane < KB && n0 + KB + lane < N) ? bk[n0 + KB + lane] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        if (n0 + k < N) {
          double part = 0.0;
          if (h0) { f0 = cur.p0[k] * (f0 + cur.w0[k] * xm1); part += cur.u0[k] * f0; }
          if (h1) { f1 = cur.p1[k] * (f1 + cur.w1[k] * xm1); part += cur.u1[k] * f1; }
          xm1 = clr::lane_value(cb, k) - wave_sum(part);
          if (lane == 0) xk[n0 + k] = xm1;
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();

  // backward with the /D of :249 folded in, cholesky.h:251-259: step n (N - 2 .. 0) reads factor row n
  f0 = 0.0;
  f1 = 0.0;
  double value = xk[N - 1] / D[N - 1];
  if (lane == 0) xk[N - 1] = value;
  {
    FactorRows<KB> nx, cur;
    double xt = 0.0, dt = 1.0;
    auto scalars = [&](long top) {  // x and D of steps top, top - 1, ...
      const long n = top - lane;
      xt = (lane < KB && n >= 0) ? xk[n] : 0.0;
      dt = (lane < KB && n >= 0) ? D[n] : 1.0;
    };
    nx.fetch(phi, u, W, J, (long)N - 2, -1, 0, (long)N - 2, j0, j1, h0, h1);
    scalars((long)N - 2);
    for (long top = (long)N - 2; top >= 0; top -= KB) {
      cur = nx;
      const double cx = xt, cd = dt;
      if (top - KB >= 0) {
        nx.fetch(phi, u, W, J, top - KB, -1, 0, (long)N - 2, j0, j1, h0, h1);
        scalars(top - KB);
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const long n = top - k;
        if (n >= 0) {
          double part = 0.0;
          if (h0) { f0 = cur.p0[k] * (f0 + cur.u0[k] * value); part += cur.w0[k] * f0; }
          if (h1) { f1 = cur.p1[k] * (f1 + cur.u1[k] * value); part += cur.w1[k] * f1; }
          value = clr::lane_value(cx, k) / clr::lane_value(cd, k) - wave_sum(part);
          if (lane == 0) xk[n] = value;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(64) dot_L_kernel(int N, int J, const double* phi,
                                                   const double* u, const double* W,
                                                   const double* D, const double* z, double* y) {
  constexpr int KB = 8;
  const int lane = threadIdx.x, j0 = lane, j1 = lane + 64;
  const bool h0 = j0 < J, h1 = j1 < J;
  const double* zk = z + (long)blockIdx.x * N;
  double* yk = y + (long)blockIdx.x * N;
  double f0 = 0.0, f1 = 0.0;
  double tmp = zk[0] * sqrt(D[0]);  // cholesky.h:421-422
  if (lane == 0) yk[0] = tmp;
  FactorRows<KB> nx, cur;
  double zt = 0.0, dt = 0.0;
  nx.fetch(phi, u, W, J, 0, +1, 0, (long)N - 2, j0, j1, h0, h1);
  zt = (lane < KB && 1 + lane < N) ? zk[1 + lane] : 0.0;
  dt = (lane < KB && 1 + lane < N) ? D[1 + lane] : 0.0;
  for (int n0 = 1; n0 < N; n0 += KB) {  // :423-427
    cur = nx;
    const double cz = zt, cd = dt;
    if (n0 + KB < N) {
      nx.fetch(phi, u, W, J, (long)n0 + KB - 1, +1, 0, (long)N - 2, j0, j1, h0, h1);
      zt = (lane < KB && n0 + KB + lane < N) ? zk[n0 + KB + lane] : 0.0;
      dt = (lane < KB && n0 + KB + lane < N) ? D[n0 + KB + lane] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (n0 + k < N) {
        double part = 0.0;
        if (h0) { f0 = cur.p0[k] * (f0 + cur.w0[k] * tmp); part += cur.u0[k] * f0; }
        if (h1) { f1 = cur.p1[k] * (f1 + cur.w1[k] * tmp); part += cur.u1[k] * f1; }
        tmp = sqrt(clr::lane_value(cd, k)) * clr::lane_value(cz, k);
        const double yn = tmp + wave_sum(part);
        if (lane == 0) yk[n0 + k] = yn;
      }
    }
  }
}

// phi, u, v of `dot` (cholesky.h:487-531): parallel over samples (blockIdx) and rows.
__global__ void __launch_bounds__(128) dot_setup_kernel(GenericProblem g, double* phi, double* u,
                                                        double* v) {
  const int J = g.J, N = g.N;
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
      const bool general = j >= g.J_real + 2 * g.J_comp;
      double ph, uu, vv;
      if (!general) {
        row_features(g, j, n, g.t[n], 0.0, ph, uu, vv);
        v[(long)J * n + j] = vv;  // v at t_n (:492-499, :505, :517-518)
        if (n >= 1) {             // u at t_n and the decay n-1 -> n are column n-1
          row_features(g, j, n, g.t[n], g.t[n] - g.t[n - 1], ph, uu, vv);
          phi[(long)J * (n - 1) + j] = ph;
          u[(long)J * (n - 1) + j] = uu;
        }
      } else {
        const int jg = j - g.J_real - 2 * g.J_comp;  // :526-530
        if (n < N - 1) {
          v[(long)J * n + j] = g.V[(long)jg * N + n];
          u[(long)J * n + j] = g.U[(long)jg * N + n + 1];
          phi[(long)J * n + j] = 1.0;
        } else {
          v[(long)J * n + j] = 0.0;  // never read by the sweeps
        }
      }
    }
  }
}

__global__ void __launch_bounds__(64) dot_kernel(int N, int J, const double* phi, const double* u,
                                                 const double* v, const double* dg,
                                                 const double* z, double* y) {
  const int j0 = threadIdx.x, j1 = threadIdx.x + 64;
  const bool h0 = j0 < J, h1 = j1 < J;
  const double* zk = z + (long)blockIdx.x * N;
  double* yk = y + (long)blockIdx.x * N;
  // upper triangle, cholesky.h:536-547
  double f0 = 0.0, f1 = 0.0;
  if (threadIdx.x == 0) yk[N - 1] = dg[N - 1] * zk[N - 1];
  for (int n = N - 2; n >= 0; --n) {
    const long base = (long)J * n;
    const double z0 = zk[n + 1];
    double part = 0.0;
    if (h0) { f0 = phi[base + j0] * (f0 + u[base + j0] * z0); part += v[base + j0] * f0; }
    if (h1) { f1 = phi[base + j1] * (f1 + u[base + j1] * z0); part += v[base + j1] * f1; }
    const double y0 = dg[n] * zk[n] + wave_sum(part);
    if (threadIdx.x == 0) yk[n] = y0;
  }
  __threadfence_block();
  __syncthreads();
  // lower triangle, :549-559
  f0 = 0.0;
  f1 = 0.0;
  for (int n = 1; n < N; ++n) {
    const long base = (long)J * (n - 1);
    const double z0 = zk[n - 1];
    double part = 0.0;
    if (h0) { f0 = phi[base + j0] * (f0 + v[base + j0] * z0); part += u[base + j0] * f0; }
    if (h1) { f1 = phi[base + j1] * (f1 + v[base + j1] * z0); part += u[base + j1] * f1; }
    const double y0 = yk[n] + wave_sum(part);
    if (threadIdx.x == 0) yk[n] = y0;
  }
}

// cholesky.h:599-698.  Row types: real rows carry one Q, a complex pair carries
// the cos-like and sin-like Q; general rows do not take part (as in the reference).
template <int R>  // rows per lane: 2 up to width 128, 16 up to 1024 (round 6)
__global__ void __launch_bounds__(64) predict_kernel(GenericProblem g, const double* alpha,
                                                     int M, const double* xs, double* pred) {
  const int N = g.N, Jrc = g.J_real + 2 * g.J_comp;
  const double* t_ = g.t;
  // each lane serves rows lane, lane + 64, ...
  double a[R], b[R], c[R], d[R];
  int kind[R];  // 0 real, 1 complex-cos row, 2 complex-sin row
#pragma unroll
  for (int r = 0; r < R; ++r) {
    a[r] = 0.0; b[r] = 0.0; c[r] = 0.0; d[r] = 0.0; kind[r] = -1;
    const int j = threadIdx.x + 64 * r;
    if (j < g.J_real) {
      kind[r] = 0; a[r] = g.a_real[j]; c[r] = g.c_real[j];
    } else if (j < Jrc) {
      const int jj = (j - g.J_real) >> 1;
      kind[r] = 1 + ((j - g.J_real) & 1);
      a[r] = g.a_comp[jj]; b[r] = g.b_comp[jj]; c[r] = g.c_comp[jj]; d[r] = g.d_comp[jj];
    }
  }
  double Q[R];
#pragma unroll
  for (int r = 0; r < R; ++r) Q[r] = 0.0;

  // forward pass :615-653
  int m = 0;
  while (m < M && xs[m] <= t_[0]) ++m;
  for (int n = 0; n < N; ++n) {
    const double alphan = alpha[n];
    const double tref = (n < N - 1) ? t_[n + 1] : t_[N - 1];
    const double tn = t_[n];
    double dt = tref - tn;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (kind[r] == 0) {
        Q[r] = (Q[r] + alphan) * exp(-c[r] * dt);
      } else if (kind[r] > 0) {
        const double tmp = exp(-c[r] * dt);
        const double tr = (kind[r] == 1) ? cos(d[r] * tn) : sin(d[r] * tn);
        Q[r] = (Q[r] + alphan * tr) * tmp;
      }
    }
    while (m < M && (n == N - 1 || xs[m] <= tref)) {
      const double xm = xs[m];
      dt = xm - tref;
      double pm = 0.0;
  #pragma unroll
    for (int r = 0; r < R; ++r) {
        if (kind[r] == 0) {
          pm += a[r] * exp(-c[r] * dt) * Q[r];
        } else if (kind[r] > 0) {
          double sd, cd;
          sincos(d[r] * xm, &sd, &cd);
          const double tmp = exp(-c[r] * dt);
          const double coef = (kind[r] == 1) ? (a[r] * cd + b[r] * sd) : (a[r] * sd - b[r] * cd);
          pm += coef * tmp * Q[r];
        }
      }
      pm = wave_sum(pm);
      if (threadIdx.x == 0) pred[m] = pm;
      ++m;
    }
  }
  __threadfence_block();
  __syncthreads();

  // backward pass :656-695
  m = M - 1;
  while (m >= 0 && xs[m] > t_[N - 1]) --m;
#pragma unroll
  for (int r = 0; r < R; ++r) Q[r] = 0.0;
  for (int n = N - 1; n >= 0; --n) {
    const double alphan = alpha[n];
    const double tref = (n > 0) ? t_[n - 1] : t_[0];
    const double tn = t_[n];
    double dt = tn - tref;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (kind[r] == 0) {
        Q[r] = (Q[r] + alphan * a[r]) * exp(-c[r] * dt);
      } else if (kind[r] > 0) {
        double sd, cd;
        sincos(d[r] * tn, &sd, &cd);
        const double coef = (kind[r] == 1) ? (a[r] * cd + b[r] * sd) : (a[r] * sd - b[r] * cd);
        Q[r] = (Q[r] + alphan * coef) * exp(-c[r] * dt);
      }
    }
    while (m >= 0 && (n == 0 || xs[m] > tref)) {
      const double xm = xs[m];
      dt = tref - xm;
      double pm = 0.0;
  #pragma unroll
    for (int r = 0; r < R; ++r) {
        if (kind[r] == 0) {
          pm += exp(-c[r] * dt) * Q[r];
        } else if (kind[r] > 0) {
          const double tmp = exp(-c[r] * dt);
          const double tr = (kind[r] == 1) ? cos(d[r] * xm) : sin(d[r] * xm);
          pm += tr * tmp * Q[r];
        }
      }
      pm = wave_sum(pm);
      if (threadIdx.x == 0) pred[m] += pm;
      --m;
    }
  }
}

}  // namespace

void launch_factor_generic(const GenericProblem& g, double* phi, double* u, double* W, double* D,
                           int* status, double* log_det, hipStream_t s) {
  const size_t lds = sizeof(double) * ((size_t)g.J * g.J + 6 * (size_t)g.J + 4);
  // (the attribute is per device: set it on every launch that needs more than the
  //  64 KB default -- J >= ~90 -- instead of caching a process-wide flag)
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_generic_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = g.J <= 8 ? 64 : 256;
  hipLaunchKernelGGL(factor_generic_kernel, dim3(1), dim3(threads), lds, s, g, phi, u, W, D,
                     status, log_det);
}

void launch_generic_loglike_batch(const GenericBatch& G, hipStream_t s) {
  const size_t J = (size_t)G.J_real + 2 * (size_t)G.J_comp + (size_t)G.J_general;
  const size_t lds = sizeof(double) * (J * J + 8 * J + 4);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_loglike_batch_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = J <= 8 ? 64 : 256;
  hipLaunchKernelGGL(generic_loglike_batch_kernel, dim3(G.B), dim3(threads), lds, s, G);
}

void launch_diag_only(int N, const double* diag, double jitter, double* D, double* log_det,
                      hipStream_t s) {
  hipLaunchKernelGGL(diag_only_kernel, dim3(1), dim3(256), 0, s, N, diag, jitter, D, log_det);
}

void launch_dot_solve(int N, int J, const double* phi, const double* u, const double* W,
                      const double* D, const double* b, double* out, hipStream_t s) {
  hipLaunchKernelGGL(dot_solve_kernel, dim3(1), dim3(64), 0, s, N, J, phi, u, W, D, b, out);
}

void launch_solve(int N, int J, int nrhs, const double* phi, const double* u, const double* W,
                  const double* D, const double* b, double* x, hipStream_t s) {
  hipLaunchKernelGGL(solve_kernel, dim3(nrhs), dim3(64), 0, s, N, J, phi, u, W, D, b, x);
}

void launch_dot_L(int N, int J, int nrhs, const double* phi, const double* u, const double* W,
                  const double* D, const double* z, double* y, hipStream_t s) {
  hipLaunchKernelGGL(dot_L_kernel, dim3(nrhs), dim3(64), 0, s, N, J, phi, u, W, D, z, y);
}

void launch_dot_setup(const GenericProblem& g, double* phi, double* u, double* v, hipStream_t s) {
  const int blocks = g.N < 4096 ? g.N : 4096;
  hipLaunchKernelGGL(dot_setup_kernel, dim3(blocks), dim3(128), 0, s, g, phi, u, v);
}

void launch_dot(int N, int J, int nrhs, const double* phi, const double* u, const double* v,
                const double* dg, const double* z, double* y, hipStream_t s) {
  hipLaunchKernelGGL(dot_kernel, dim3(nrhs), dim3(64), 0, s, N, J, phi, u, v, dg, z, y);
}

void launch_predict(const GenericProblem& g, const double* alpha, int M, const double* xs,
                    double* pred, hipStream_t s) {
  if (g.J_real + 2 * g.J_comp <= 128) hipLaunchKernelGGL((predict_kernel<2>), dim3(1), dim3(64), 0, s, g, alpha, M, xs, pred);
  else hipLaunchKernelGGL((predict_kernel<16>), dim3(1), dim3(64), 0, s, g, alpha, M, xs, pred);
}

}  // namespace clr
