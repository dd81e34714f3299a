// celerite_amd/csrc/small_kernels.hip -- CholeskySolver.compute of ONE short series (N <= 4096, widths 1..4) in ONE
// launch (BASELINE configs[0]: N = 1000, a real + an SHO term, through the object API).
//
// The general object-API route (api.hip: summarize -> prefix -> correct -> replay -> sequential -> finalize) is seven
// launches and five uploads for one problem: ~100 us of device time and as much host time at N = 1000, where a CPU core
// needs 55 us.  Here the whole factorisation is one workgroup:
//   1. lane = chunk of L = ceil(N / T) samples: summarize_chunk (clr_core.h) folds the chunk into its transfer
//      element, registers only;
//   2. a Kogge-Stone inclusive scan over the lanes with the element composition (compose_elements, clr_core.h) through
//      LDS: log2(T) levels; lane c then holds the element of chunks 0..c, whose (C, b) -- the zero-start trajectory --
//      IS the state at the first sample of chunk c + 1: no advance step, no Gauss-Jordan beyond the compositions;
//   3. replay_chunk from that state: the reference recurrence (cholesky.h:126-179), writing phi, u, W, D in the
//      reference's storage and the chunk's log-det / quadratic sums;
//   4. every chunk's end state is compared with the next chunk's start state (the same check as the batched replay,
//      clr_batch_kernels.h): consistent everywhere and no pivot flagged => done; anything else (mismatch above 1e-11,
//      a non-positive pivot, a non-finite number) => status "fallback" and the host takes the general route.
// Coefficients and jitter travel as kernel arguments; t, diag and the hinted right-hand side in one upload.
#include <hip/hip_runtime.h>

#include "clr_core.h"
#include "clr_small.h"

namespace clr {

namespace {

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(256) small_compute_kernel(const SmallParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J, SZ = Wd::SZ, ELEM = Wd::ELEM, START = Wd::START;
  extern __shared__ double lds[];  // [ELEM][T] elements of the scan; reused for the end-state check and reductions
  const int T = blockDim.x, c = threadIdx.x;
  const int L = P.L, N = P.N;
  const int nreal = (N + L - 1) / L;  // chunks that hold samples
  Problem<JR, JC> p;
  p.load(P.coeff, P.coeff + JR, P.coeff + 2 * JR, P.coeff + 2 * JR + JC, P.coeff + 2 * JR + 2 * JC,
         P.coeff + 2 * JR + 3 * JC, P.jitter);
  const double* yv = P.y ? P.y : P.t;  // (without a right-hand side the quadratic sums are never looked at)
  auto series = [&]() {
    return DirectSeries{P.t + (long)c * L, P.diag + (long)c * L, yv + (long)c * L, 1, L, L, (long)N - (long)c * L};
  };

  // 1. the chunk's element
  double e[ELEM];
  {
    double ld0, q0;
    int flag0;
    DirectSeries src = series();
    summarize_chunk<JR, JC, FAST>(p, src, L, c * L, N, true, e, &ld0, &q0, &flag0);
  }
  // 2. inclusive scan over the lanes (chunk order) with the composition
  for (int d = 1; d < nreal; d <<= 1) {
#pragma unroll
    for (int k = 0; k < ELEM; ++k) lds[k * T + c] = e[k];
    __syncthreads();
    if (c >= d && c < nreal) {
      double left[ELEM];
#pragma unroll
      for (int k = 0; k < ELEM; ++k) left[k] = lds[k * T + c - d];
      compose_elements<J>(left, e, e);
    }
    __syncthreads();
  }
  // the state at the first sample of chunk c = (C, b) of the element of chunks 0 .. c - 1
#pragma unroll
  for (int k = 0; k < SZ; ++k) lds[k * T + c] = e[J * J + J + k];
#pragma unroll
  for (int k = 0; k < J; ++k) lds[(SZ + k) * T + c] = e[J * J + k];
  __syncthreads();
  double start[START];
#pragma unroll
  for (int k = 0; k < START; ++k) start[k] = (c > 0) ? lds[k * T + c - 1] : 0.0;
  __syncthreads();

  // 3. the reference recurrence from that state, writing the factor
  double ld, qd, endst[START];
  int flag;
  {
    DirectSeries src = series();
    replay_chunk<JR, JC, 1, FAST>(p, src, L, N, c * L, c > 0 ? start : nullptr, &ld, &qd, &flag, P.phi, P.u, P.W, P.D, 0,
                                  endst);
  }
  // 4. boundary check: this chunk's end state against the next chunk's start state
#pragma unroll
  for (int k = 0; k < START; ++k) lds[k * T + c] = endst[k];
  __syncthreads();
  double res = 0.0;
  int bad = 0;
  if (c >= 1 && c < nreal) {
    double pm = 0.0, dp = 0.0, fm = 0.0, df = 0.0;
#pragma unroll
    for (int k = 0; k < SZ; ++k) {
      const double en = lds[k * T + c - 1];
      pm = fmax(pm, fabs(en));
      dp = fmax(dp, fabs(en - start[k]));
    }
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const double en = lds[(SZ + k) * T + c - 1];
      fm = fmax(fm, fabs(en));
      df = fmax(df, fabs(en - start[SZ + k]));
    }
    res = (pm > 0.0) ? dp / pm : (dp == 0.0 ? 0.0 : INFINITY);
    if (P.y && fm > 0.0) res = fmax(res, df / fm);
    if (!(res <= P.max_residual)) bad = 1;  // (NaN counts as inconsistent)
  }
  if (c < nreal && (flag || !isfinite(ld) || (P.y && !isfinite(qd)))) bad = 1;
  __syncthreads();
  // sums in chunk order: a fixed tree over the lanes (lanes past the last chunk contribute zeros)
  lds[c] = (c < nreal) ? ld : 0.0;
  lds[T + c] = (c < nreal) ? qd : 0.0;
  lds[2 * T + c] = (double)bad;
  lds[3 * T + c] = (res == res) ? res : INFINITY;
  __syncthreads();
  for (int s = T / 2; s >= 1; s >>= 1) {
    if (c < s) {
      lds[c] += lds[c + s];
      lds[T + c] += lds[T + c + s];
      lds[2 * T + c] += lds[2 * T + c + s];
      lds[3 * T + c] = fmax(lds[3 * T + c], lds[3 * T + c + s]);
    }
    __syncthreads();
  }
  if (c == 0) {
    P.out[0] = (lds[2 * T] > 0.0) ? -1.0 : 0.0;  // -1: the general route must settle this problem
    P.out[1] = lds[0];
    P.out[2] = lds[T];
    P.out[3] = lds[3 * T];
  }
}

template <int JR, int JC>
bool go(const SmallParams& P, int threads, bool fast, hipStream_t s) {
  constexpr int ELEM = Widths<JR, JC>::ELEM;
  const size_t lds = (size_t)(ELEM > 4 ? ELEM : 4) * threads * sizeof(double);
  if (fast) hipLaunchKernelGGL((small_compute_kernel<JR, JC, true>), dim3(1), dim3(threads), lds, s, P);
  else hipLaunchKernelGGL((small_compute_kernel<JR, JC, false>), dim3(1), dim3(threads), lds, s, P);
  return true;
}

}  // namespace

bool small_compute_supported(int JR, int JC, int N) {
  const int J = JR + 2 * JC;
  return J >= 1 && J <= 4 && N >= 64 && N <= 4096;
}

bool launch_small_compute(int JR, int JC, const SmallParams& P, int threads, bool fast, hipStream_t s) {
#define CLR_SMALL(R, C) if (JR == R && JC == C) return go<R, C>(P, threads, fast, s);
  CLR_SMALL(1, 0) CLR_SMALL(2, 0) CLR_SMALL(3, 0) CLR_SMALL(4, 0)
  CLR_SMALL(0, 1) CLR_SMALL(1, 1) CLR_SMALL(2, 1) CLR_SMALL(0, 2)
#undef CLR_SMALL
  return false;
}

}  // namespace clr
