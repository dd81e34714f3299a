// celerite_amd/csrc/huge_kernels.hip -- CholeskySolver.compute / dot_solve / solve / log_determinant at ANY width above
// 128 (to CLR_MAX_WIDTH_ANY): the reference's dynamic-width arm (cholesky.h:203, `FIXED_SIZE_HACKZ(Eigen::Dynamic)`)
// takes any J, and its published benchmark goes to width 512 (examples/benchmark/run.py:39).
//
// Up to width 128 the any-width kernel keeps S (J^2 doubles) in LDS (generic_kernels.hip: 131 KB at 128); here S lives
// in HBM -- 2 MB at width 512: it stays in the XCD's L2 between two steps -- and ONE workgroup of 1024 threads walks the
// series exactly as the reference does (sequential in n, parallel over the J^2 entries of a step):
//   features    thread per row: phi, u~, v~ of the sample (cholesky.h:129-152) into LDS;
//   S and q     ONE pass over the upper triangle per step, a wave per column j, lanes over the rows k <= j (coalesced):
//               S[k + J j] <- phi_j (phi_k (S + D_{n-1} W_j W_k)) (cholesky.h:154-160), used at once for q = S u~
//               (cholesky.h:163-175): the column's sum by a wave reduction, the mirrored contributions q_k += S_kj u_j in
//               the lane's own registers; every wave's share of q in its own LDS row, summed in wave order (no atomics:
//               the result does not depend on scheduling); then D_n = d_n - u~ . q, W_n = (v~ - q) / D_n.
// The step is bound by ONE compute unit's path to L2 (16 B per triangle entry): see profiles/r06o_huge_width.txt.
// The sweeps over the stored factor keep f (J doubles) in LDS, thread per row, one workgroup per right-hand side.
// Not here (CLR_UNSUPPORTED above 128): dot_L, dot, predict, grad_log_likelihood, the batched plans.
#include <hip/hip_runtime.h>

#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"

namespace clr {

namespace {

__device__ __forceinline__ double hwave_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// phi, u~, v~ of row j at sample n (cholesky.h:129-152); the same arithmetic as generic_kernels.hip's row_features
__device__ __forceinline__ void huge_row_features(const GenericProblem& g, int j, int n, double t, double dx,
                                                  double& phi, double& u, double& v) {
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

// block-wide sum of one value per thread (blockDim a multiple of 64, <= 1024); red: 16 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = hwave_sum(v);
  __syncthreads();  // (red may still be read from a previous call)
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += red[w];  // (every thread: the same order, the same sum)
  return s;
}

constexpr int HUGE_WAVES = 8;     // waves of the factorisation's workgroup

// ROWS = rows per lane = ceil(J / 64) rounded up to 4 / 8 / 16; UC = columns of a group (32 entries per lane in flight)
template <int ROWS>
__global__ void __launch_bounds__(64 * HUGE_WAVES) factor_huge_kernel(GenericProblem g, double* __restrict__ S /* [J][J]: column j holds rows k <= j */,
                                                                       double* phi, double* u, double* W, double* D, int* status,
                                                                       double* log_det) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int J = g.J, N = g.N, tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  constexpr int UC = 32 / ROWS;
  double* sphi = reinterpret_cast<double*>(smem);
  double* su = sphi + J;
  double* sv = su + J;
  double* swp = sv + J;
  double* sq = swp + J;
  double* red = sq + J;       // 16
  double* partial = red + 16;  // [nw][J]: every wave's share of q = S u~ (summed in wave order: deterministic)
  double* mine = partial + (long)wave * J;
  // the rows' coefficients, read once: (a | c) of a real row, (a, b, c, d) of a complex pair's rows
  double* ca = partial + (long)nw * J;
  double* cb = ca + J;
  double* cc = cb + J;
  double* cd = cc + J;
  const int Jc = g.J_real + 2 * g.J_comp;
  for (int j = tid; j < Jc; j += nt) {
    if (j < g.J_real) { ca[j] = g.a_real[j]; cb[j] = 0.0; cc[j] = g.c_real[j]; cd[j] = 0.0; }
    else { const int jj = (j - g.J_real) >> 1; ca[j] = g.a_comp[jj]; cb[j] = g.b_comp[jj]; cc[j] = g.c_comp[jj]; cd[j] = g.d_comp[jj]; }
  }
  const long JJ = (long)J * J;
  for (long i = tid; i < JJ; i += nt) S[i] = 0.0;  // cholesky.h:124
  __syncthreads();
  // phi, u~, v~ of row j at sample n (cholesky.h:129-152) from the cached coefficients
  auto features = [&](int j, int n, double t, double dx, double& ph, double& uu, double& vv) {
    if (j < g.J_real) {
      ph = exp(-cc[j] * dx); uu = ca[j]; vv = 1.0;
    } else if (j < Jc) {
      const bool odd = (j - g.J_real) & 1;
      double sd, cdv;
      sincos(cd[j] * t, &sd, &cdv);
      ph = exp(-cc[j] * dx);
      uu = odd ? (ca[j] * sd - cb[j] * cdv) : (ca[j] * cdv + cb[j] * sd);
      vv = odd ? sd : cdv;
    } else {
      const int jg = j - Jc;
      ph = 1.0; uu = g.U[(long)jg * N + n]; vv = g.V[(long)jg * N + n];
    }
  };

  // sample 0: cholesky.h:100-117
  double Dprev = D[0];
  double ld = log(Dprev);
  {
    const double value = 1.0 / Dprev;
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      features(j, 0, g.t[0], 0.0, ph, uu, vv);
      const double w = vv * value;
      W[j] = w;
      swp[j] = w;
    }
  }
  __syncthreads();

  double t_prev = g.t[0], t_cur = N > 1 ? g.t[1] : 0.0, d_cur = N > 1 ? D[1] : 0.0;
  for (int n = 1; n < N; ++n) {
    const double t = t_cur, dx = t - t_prev, d_in = d_cur;
    if (n + 1 < N) { t_cur = g.t[n + 1]; d_cur = D[n + 1]; }  // (a step ahead: off the step's critical path)
    t_prev = t;
    for (int j = tid; j < J; j += nt) {
      double ph, uu, vv;
      features(j, n, t, dx, ph, uu, vv);
      sphi[j] = ph;
      su[j] = uu;
      sv[j] = vv;
      phi[(long)J * (n - 1) + j] = ph;
      u[(long)J * (n - 1) + j] = uu;
    }
    for (int k = lane; k < J; k += 64) mine[k] = 0.0;
    __syncthreads();
    // ONE pass over the upper triangle, a wave per column (lanes over the rows k <= j, coalesced): the entry is moved to
    // sample n (cholesky.h:154-160) and used at once for both of its contributions to q = S u~ (cholesky.h:163-175):
    // q_j += S_kj u_k (the column's sum: a wave reduction) and, off the diagonal, q_k += S_kj u_j (this lane's own rows).
    double rowacc[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) rowacc[m] = 0.0;
    // (software pipeline: the NEXT group's columns are requested before this group's entries are stored -- a store to S
    //  orders every later load of S, and an L2 round trip is ~1.2 us against ~0.1 us of arithmetic per column)
    double val[UC][ROWS], nxt[UC][ROWS];
    auto load_group = [&](double (&dst)[UC][ROWS], int j0) {
#pragma unroll
      for (int c = 0; c < UC; ++c) {
        const int j = j0 + c * nw;
        const double* col = S + (long)J * j;
#pragma unroll
        for (int m = 0; m < ROWS; ++m) {
          const int k = lane + 64 * m;
          dst[c][m] = (j < J && k <= j) ? col[k] : 0.0;
        }
      }
    };
    load_group(val, wave);
    for (int j0 = wave; j0 < J; j0 += nw * UC) {
      load_group(nxt, j0 + nw * UC);
#pragma unroll
      for (int c = 0; c < UC; ++c) {
        const int j = j0 + c * nw;
        if (j < J) {  // (wave-uniform)
          double* col = S + (long)J * j;
          const double pj = sphi[j], xj = Dprev * swp[j], uj = su[j];
          double colacc = 0.0;
#pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            const int k = lane + 64 * m;
            if (k <= j) {
              const double v = pj * (sphi[k] * (val[c][m] + xj * swp[k]));
              col[k] = v;
              colacc = fma(v, su[k], colacc);
              if (k < j) rowacc[m] = fma(v, uj, rowacc[m]);
            }
          }
          colacc = hwave_sum(colacc);
          if (lane == 0) mine[j] += colacc;
        }
      }
#pragma unroll
      for (int c = 0; c < UC; ++c) {
#pragma unroll
        for (int m = 0; m < ROWS; ++m) val[c][m] = nxt[c][m];
      }
    }
    // (lane 0's column sums above and the row sums below touch the wave's OWN share only; a wave's LDS operations
    //  complete in order, so no fence is needed between them)
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int k = lane + 64 * m;
      if (k < J) mine[k] += rowacc[m];
    }
    __syncthreads();
    double part = 0.0;
    for (int j = tid; j < J; j += nt) {
      double q = 0.0;
      for (int w = 0; w < nw; ++w) q += partial[(long)w * J + j];
      sq[j] = q;
      part = fma(su[j], q, part);
    }
    const double total = block_sum(part, red);
    const double Dn = d_in - total;
    if (Dn < 0.0) {  // cholesky.h:176 (every thread sees the same Dn)
      if (tid == 0) { status[0] = 1; log_det[0] = NAN; }
      return;
    }
    if (tid == 0) D[n] = Dn;
    ld += log(Dn);
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

// dot_solve (cholesky.h:343-357) and solve (:236-260) over a stored factor, one workgroup per right-hand side,
// thread per row (strided), f in LDS.  x (solve) is column-major [N][nrhs] like b.
template <bool SOLVE>
__global__ void __launch_bounds__(256) sweep_huge_kernel(int N, int J, const double* __restrict__ phi,
                                                         const double* __restrict__ u, const double* __restrict__ W,
                                                         const double* __restrict__ D, const double* __restrict__ b,
                                                         double* __restrict__ x, double* quad_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* f = reinterpret_cast<double*>(smem);
  double* red = f + J;
  const int tid = threadIdx.x, nt = blockDim.x, rhs = blockIdx.x;
  const double* bb = b + (long)rhs * N;
  double* xx = SOLVE ? x + (long)rhs * N : nullptr;
  for (int j = tid; j < J; j += nt) f[j] = 0.0;
  double xm1 = bb[0];
  double result = xm1 * (xm1 / D[0]);  // :347
  if (SOLVE && tid == 0) xx[0] = xm1;
  __syncthreads();
  for (int n = 1; n < N; ++n) {  // forward: :240-248 / :350-354
    const long base = (long)J * (n - 1);
    double part = 0.0;
    for (int j = tid; j < J; j += nt) {
      const double value = phi[base + j] * (f[j] + W[base + j] * xm1);
      f[j] = value;
      part += u[base + j] * value;
    }
    const double xv = bb[n] - block_sum(part, red);
    xm1 = xv;
    result += xv * xv / D[n];  // :356
    if (SOLVE && tid == 0) xx[n] = xv;
  }
  if (!SOLVE) {
    if (tid == 0) quad_out[rhs] = result;
    return;
  }
  __syncthreads();
  for (int n = tid; n < N; n += nt) xx[n] /= D[n];  // :249
  for (int j = tid; j < J; j += nt) f[j] = 0.0;
  __syncthreads();
  double xnp1 = xx[N - 1];
  for (int n = N - 2; n >= 0; --n) {  // backward: :252-259
    const long base = (long)J * n;
    double part = 0.0;
    for (int j = tid; j < J; j += nt) {
      const double value = phi[base + j] * (f[j] + u[base + j] * xnp1);
      f[j] = value;
      part += W[base + j] * value;
    }
    const double x_in = xx[n];  // (read before the barriers of the sum: thread 0 overwrites it below)
    const double xv = x_in - block_sum(part, red);
    xnp1 = xv;
    if (tid == 0) xx[n] = xv;
  }
}

}  // namespace

size_t factor_huge_workspace_doubles(int J) { return (size_t)J * (size_t)J; }

void launch_factor_huge(const GenericProblem& g, double* S, double* phi, double* u, double* W, double* D, int* status,
                        double* log_det, hipStream_t s) {
  const size_t lds = sizeof(double) * ((9 + HUGE_WAVES) * (size_t)g.J + 16);  // 139 KB at width 1024
  const dim3 grid(1), block(64 * HUGE_WAVES);
#define CLR_HUGE_GO(R)                                                                                                       \
  do {                                                                                                                       \
    if (lds > 64 * 1024)                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_huge_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((factor_huge_kernel<R>), grid, block, lds, s, g, S, phi, u, W, D, status, log_det);                   \
  } while (0)
  if (g.J <= 256) CLR_HUGE_GO(4);
  else if (g.J <= 512) CLR_HUGE_GO(8);
  else CLR_HUGE_GO(16);
#undef CLR_HUGE_GO
}

void launch_dot_solve_huge(int N, int J, const double* phi, const double* u, const double* W, const double* D,
                           const double* b, double* out, hipStream_t s) {
  const size_t lds = sizeof(double) * ((size_t)J + 16);
  hipLaunchKernelGGL((sweep_huge_kernel<false>), dim3(1), dim3(256), lds, s, N, J, phi, u, W, D, b, nullptr, out);
}

void launch_solve_huge(int N, int J, int nrhs, const double* phi, const double* u, const double* W, const double* D,
                       const double* b, double* x, hipStream_t s) {
  const size_t lds = sizeof(double) * ((size_t)J + 16);
  hipLaunchKernelGGL((sweep_huge_kernel<true>), dim3(nrhs), dim3(256), lds, s, N, J, phi, u, W, D, b, x, nullptr);
}

}  // namespace clr
