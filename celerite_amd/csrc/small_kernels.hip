// celerite_amd/csrc/small_kernels.hip -- CholeskySolver.compute of ONE short series (N <= 4096, widths 1..4) in ONE
// launch (BASELINE configs[0]: N = 1000, a real + an SHO term, through the object API).
//
// The general object-API route (api_solver.hip: summarize -> prefix -> correct -> replay -> sequential -> finalize) is seven
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

#include "../../include/celerite_hip.h"
#include "clr_batch_kernels.h"
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

// ---------------------------------------------------------------------------------------------------------------
// The same idea for a BATCH of short problems (BASELINE configs[1]: 256 problems x N = 1e4 x width 4): ONE workgroup
// per problem, one launch for the whole fused log-likelihood instead of the six launches of the scan pipeline
// (summarize, prefix, correct, replay, sequential, finalize: 0.17 ms of device time at that shape, the prefix over 250
// chunks longer than the summarize).  Lane = chunk of L = ceil(N / T) samples: (1) summarize_chunk, (2) Kogge-Stone scan
// of the composed elements through LDS -- (C, b) of the element of chunks 0 .. c - 1 IS the state at chunk c's first
// sample --, (3) the chunk's true contributions from its zero-start sums and that state (chunk_update: determinant
// lemma, Woodbury, positivity certificate -- what correct_kernel does; a replay of the chunk instead was measured at
// 41 of 117 us), (4) the problem's conditioning record tested as decide_kernel does.  Certified and benign => the
// problem's results are written by this kernel; anything else => status "pending" and need_scan[b] = 1: the host
// runs the scan pipeline (with all its routes) for those problems before it hands results out, as behind the
// warm-started recurrence (clr_batch_kernels.h).
// ---------------------------------------------------------------------------------------------------------------
// DENSE: the plan's series are densely sampled (BatchParams::dense, lazy_eligible on the host): the complex terms'
// phases advance by small-angle rotations (summarize_chunk, DENSE) -- the summarize phase is issue-bound (one wave per
// SIMD at one instruction per ~2.4 ns), so its instruction count is its time.
template <int JR, int JC, bool FAST, bool DENSE>
__global__ void __launch_bounds__(256) small_batch_kernel(const BatchParams P, int L) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J, SZ = Wd::SZ, ELEM = Wd::ELEM;
  extern __shared__ double lds[];  // [max(ELEM, 8)][T]
  const int T = blockDim.x, c = threadIdx.x, b = blockIdx.x;
  const int N = P.N;
  const int nreal = (N + L - 1) / L;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  // 1. the chunk's element and zero-start sums; eta and Jm of the chunk's OWN element are what the corrections need
  double e[ELEM], own[ELEM];
  double ld0 = 0.0, q0 = 0.0, gamma = 0.0;
  int flag0 = 0;
  {
    // (the chunk-interleaved copy: lane_is = 256, lane_cs = 1 -- api_batch.hip: small_params)
    DirectSeries src{P.t + (long)b * P.t_stride + (long)c * P.lane_cs, P.diag + (long)b * P.diag_stride + (long)c * P.lane_cs,
                     P.y + (long)b * P.y_stride + (long)c * P.lane_cs, P.lane_is, P.lane_cs, L, (long)N - (long)c * L};
    summarize_chunk<JR, JC, FAST, DirectSeries, DENSE>(p, src, L, c * L, N, true, e, &ld0, &q0, &flag0, &gamma);
  }
#pragma unroll
  for (int k = 0; k < ELEM; ++k) own[k] = (k >= J * J + J + SZ) ? e[k] : 0.0;  // (eta | Jm; the rest is never read)
  // 2. inclusive Kogge-Stone scan over the lanes (chunk order) with the composition
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
  double S[SZ], f[J];
#pragma unroll
  for (int k = 0; k < SZ; ++k) S[k] = (c > 0) ? lds[k * T + c - 1] : 0.0;
#pragma unroll
  for (int k = 0; k < J; ++k) f[k] = (c > 0) ? lds[(SZ + k) * T + c - 1] : 0.0;
  __syncthreads();
  // 3. the chunk's true contributions from its zero-start sums and its start state (chunk_update, clr_core.h:
  //    determinant lemma + Woodbury + the positivity certificate), as correct_kernel does -- no second pass
  double dld = 0.0, dq = 0.0, mu = 1.0, eg = 0.0, errq = 0.0;
  int sus = 0;
  if (c >= 1 && c < nreal) chunk_update<J>(own, S, f, true, false, ld0, q0, &dld, &dq, &sus, &mu, true, &eg, &errq);
  const double ld = ld0 + dld, qd = q0 + dq;
  int bad = 0;
  if (c < nreal && (flag0 || sus || !isfinite(ld) || !isfinite(qd))) bad = 1;
  // 4. sums in chunk order (a fixed tree over the lanes) and the problem's conditioning record (decide_kernel's test)
  auto nmax = [](double a, double x) { return (a != a) ? a : ((x != x) ? x : (x > a ? x : a)); };
  auto nmin = [](double a, double x) { return (a != a) ? a : ((x != x) ? x : (x < a ? x : a)); };
  const bool real = c < nreal;
  lds[c] = real ? ld : 0.0;
  lds[T + c] = real ? qd : 0.0;
  lds[2 * T + c] = (double)bad;
  lds[3 * T + c] = real ? gamma : 0.0;
  lds[4 * T + c] = real ? mu : 1.0;
  lds[5 * T + c] = real ? eg : 0.0;
  lds[6 * T + c] = (real && c >= 1) ? J * 2.2e-16 / mu : 0.0;  // the corrections' rounding-error estimates, summed over the problem (decide_kernel)
  lds[7 * T + c] = real ? errq : 0.0;
  __syncthreads();
  for (int s = T / 2; s >= 1; s >>= 1) {
    if (c < s) {
      lds[c] += lds[c + s];
      lds[T + c] += lds[T + c + s];
      lds[2 * T + c] += lds[2 * T + c + s];
      lds[3 * T + c] = nmax(lds[3 * T + c], lds[3 * T + c + s]);
      lds[4 * T + c] = nmin(lds[4 * T + c], lds[4 * T + c + s]);
      lds[5 * T + c] = nmax(lds[5 * T + c], lds[5 * T + c + s]);
      lds[6 * T + c] += lds[6 * T + c + s];
      lds[7 * T + c] += lds[7 * T + c + s];
    }
    __syncthreads();
  }
  if (c == 0) {
    const double g = lds[3 * T], m = lds[4 * T], er = lds[5 * T];
    bool pending = lds[2 * T] > 0.0;
    if (!(lds[6 * T] <= 3e-12 * fabs(lds[0])) || !(lds[7 * T] <= 3e-12 * fabs(lds[T]))) pending = true;
    if (P.cert_gamma > 0.0 && (!(g < P.cert_gamma * m) || (P.cert_gamma_abs > 0.0 && !(g < P.cert_gamma_abs)) ||
                               (P.cert_eg > 0.0 && !(g * er < P.cert_eg))))
      pending = true;  // ill-conditioned: the scan pipeline routes it (checked replay / sequential recurrence)
    if (pending) {
      P.need_scan[b] = 1;
      P.out_status[b] = CLR_PENDING_STATUS;
    } else {
      P.need_scan[b] = 0;
      P.need_exact[b] = 0;  // (route 0 for clr_batch_get_exact_flags: settled without the reference recurrence)
      P.out_status[b] = CLR_OK;
      P.out_logdet[b] = lds[0];
      P.out_quad[b] = lds[T];
      P.out_ll[b] = combine_loglike(lds[0], lds[T], N);
    }
  }
}

template <int JR, int JC>
bool go_batch(const BatchParams& P, int threads, hipStream_t s) {
  constexpr int ELEM = Widths<JR, JC>::ELEM;
  const int L = (P.N + threads - 1) / threads;
  const size_t lds = (size_t)(ELEM > 8 ? ELEM : 8) * threads * sizeof(double);
  if (P.fast_trig) {
    if (P.dense && JC > 0) hipLaunchKernelGGL((small_batch_kernel<JR, JC, true, true>), dim3(P.B), dim3(threads), lds, s, P, L);
    else hipLaunchKernelGGL((small_batch_kernel<JR, JC, true, false>), dim3(P.B), dim3(threads), lds, s, P, L);
  } else hipLaunchKernelGGL((small_batch_kernel<JR, JC, false, false>), dim3(P.B), dim3(threads), lds, s, P, L);
  return true;
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

bool small_batch_supported(int JR, int JC, int N) {
  const int J = JR + 2 * JC;
  return J >= 1 && J <= 4 && N >= 512 && N <= 32768;
}

bool launch_small_batch(int JR, int JC, const BatchParams& P, int threads, hipStream_t s) {
#define CLR_SMALLB(R, C) if (JR == R && JC == C) return go_batch<R, C>(P, threads, s);
  CLR_SMALLB(1, 0) CLR_SMALLB(2, 0) CLR_SMALLB(3, 0) CLR_SMALLB(4, 0)
  CLR_SMALLB(0, 1) CLR_SMALLB(1, 1) CLR_SMALLB(2, 1) CLR_SMALLB(0, 2)
#undef CLR_SMALLB
  return false;
}

bool launch_small_compute(int JR, int JC, const SmallParams& P, int threads, bool fast, hipStream_t s) {
#define CLR_SMALL(R, C) if (JR == R && JC == C) return go<R, C>(P, threads, fast, s);
  CLR_SMALL(1, 0) CLR_SMALL(2, 0) CLR_SMALL(3, 0) CLR_SMALL(4, 0)
  CLR_SMALL(0, 1) CLR_SMALL(1, 1) CLR_SMALL(2, 1) CLR_SMALL(0, 2)
#undef CLR_SMALL
  return false;
}

}  // namespace clr
