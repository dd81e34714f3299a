// celerite_amd/csrc/clr_batch_kernels.h
//
// gfx950 kernels of the batched (problem x chunk) scan; see clr_core.h for the
// algebra.  Lane mapping:
//   summarize / replay : grid (ceil(nchunk/64), B), one 64-lane wave per block;
//       lane = chunk index within the problem blockIdx.y.  All 64 lanes of a wave
//       work on the SAME problem, so its hyper-parameters are wave-uniform (SGPRs,
//       scalar loads) and only the recurrence state lives in VGPRs.
//   prefix / finalize  : one lane per problem (sequential over that problem's
//       chunks; the work is O(nchunk J^3), a few percent of the total).
// HBM traffic: 24 B per sample per pass (t, diag, y).  The series are read in the
// chunk-interleaved layout [problem][i][chunk] (relayout_kernel in api.hip), so
// the 64 lanes of a wave -- 64 consecutive chunks at the same local step i -- load
// 512 contiguous bytes per array per step: every cache line is touched once.
// (With the row-major API layout each lane streams its own 8 B/step run and the
// 64-B lines must survive 8 steps in L1/L2: measured 2x slower at 2 waves/SIMD.)
// The workspace (elements, start states, partial sums) is O(nchunk) per problem.
#pragma once

#include <hip/hip_runtime.h>

#include "clr_core.h"

namespace clr {

struct BatchParams {
  int B, N, nchunk, L;
  const double *jitter, *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp;
  // series as the kernels read them (see SeriesLane): element (chunk c, local i) of
  // problem b is at  base[b * stride + c * lane_cs + i * lane_is]
  const double *t, *diag, *y;
  long t_stride, diag_stride, y_stride;
  long lane_is, lane_cs;
  int fast_trig;  // host-verified: max|d_comp| * max|t| < CLR_FAST_TRIG_LIMIT
  double* elems;   // [B][nchunk][ELEM]
  double* starts;  // [B][nchunk][START]
  double* part;    // [B][nchunk][2]  (sum log D, sum x^2/D)
  int* flags;      // [B][nchunk]
  double *out_ll, *out_logdet, *out_quad;
  int* out_status;
  // factor (reference layout), only for materialising runs
  double *phi, *u, *W, *D;
};

template <int JR, int JC>
__device__ __forceinline__ void load_problem(const BatchParams& P, int b, Problem<JR, JC>& p) {
  p.load(P.a_real + (long)b * JR, P.c_real + (long)b * JR, P.a_comp + (long)b * JC,
         P.b_comp + (long)b * JC, P.c_comp + (long)b * JC, P.d_comp + (long)b * JC,
         P.jitter[b]);
}

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) summarize_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk - 1) return;  // the last chunk's element is never needed
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  const SeriesLane sl{P.t + b * P.t_stride + c * P.lane_cs, P.diag + b * P.diag_stride + c * P.lane_cs,
                      P.y + b * P.y_stride + c * P.lane_cs, P.lane_is, P.lane_cs, P.L};
  summarize_chunk<JR, JC, FAST>(p, sl, P.elems + ((long)b * P.nchunk + c) * Wd::ELEM);
}

template <int JR, int JC>
__global__ void __launch_bounds__(64) prefix_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  double S[Wd::SZ], f[J];
#pragma unroll
  for (int i = 0; i < Wd::SZ; ++i) S[i] = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) f[i] = 0.0;
  for (int c = 0; c + 1 < P.nchunk; ++c) {
    apply_element<J>(P.elems + ((long)b * P.nchunk + c) * Wd::ELEM, S, f);
    double* o = P.starts + ((long)b * P.nchunk + c + 1) * Wd::START;
#pragma unroll
    for (int i = 0; i < Wd::SZ; ++i) o[i] = S[i];
#pragma unroll
    for (int i = 0; i < J; ++i) o[Wd::SZ + i] = f[i];
  }
}

template <int JR, int JC, bool MATERIALIZE, bool FAST>
__global__ void __launch_bounds__(64) replay_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const int n0 = c * P.L;
  if (n0 >= P.N) {  // empty trailing chunk (nchunk * L may exceed N)
    P.part[((long)b * P.nchunk + c) * 2 + 0] = 0.0;
    P.part[((long)b * P.nchunk + c) * 2 + 1] = 0.0;
    P.flags[(long)b * P.nchunk + c] = 0;
    return;
  }
  const int n1 = min(n0 + P.L, P.N);
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  double ld, qd;
  int flag;
  const long Nm1 = P.N - 1;
  const SeriesLane sl{P.t + b * P.t_stride + c * P.lane_cs, P.diag + b * P.diag_stride + c * P.lane_cs,
                      P.y + b * P.y_stride + c * P.lane_cs, P.lane_is, P.lane_cs, P.L};
  replay_chunk<JR, JC, MATERIALIZE, FAST>(
      p, sl, P.N, n0, n1,
      c == 0 ? nullptr : P.starts + ((long)b * P.nchunk + c) * Wd::START, &ld, &qd, &flag,
      MATERIALIZE ? P.phi + (long)b * J * Nm1 : nullptr,
      MATERIALIZE ? P.u + (long)b * J * Nm1 : nullptr,
      MATERIALIZE ? P.W + (long)b * J * P.N : nullptr,
      MATERIALIZE ? P.D + (long)b * P.N : nullptr);
  P.part[((long)b * P.nchunk + c) * 2 + 0] = ld;
  P.part[((long)b * P.nchunk + c) * 2 + 1] = qd;
  P.flags[(long)b * P.nchunk + c] = flag;
}

// One table entry per (JR, JC): host-callable launchers.
struct BatchLaunchers {
  void (*summarize)(const BatchParams&, hipStream_t);
  void (*prefix)(const BatchParams&, hipStream_t);
  void (*replay)(const BatchParams&, bool materialize, hipStream_t);
  int elem_doubles, start_doubles;
};

template <int JR, int JC>
struct BatchImpl {
  static void summarize(const BatchParams& P, hipStream_t s) {
    if (P.nchunk < 2) return;
    dim3 grid((P.nchunk - 1 + 63) / 64, P.B);
    if (P.fast_trig)
      hipLaunchKernelGGL((summarize_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    else
      hipLaunchKernelGGL((summarize_kernel<JR, JC, false>), grid, dim3(64), 0, s, P);
  }
  static void prefix(const BatchParams& P, hipStream_t s) {
    if (P.nchunk < 2) return;
    hipLaunchKernelGGL((prefix_kernel<JR, JC>), dim3((P.B + 63) / 64), dim3(64), 0, s, P);
  }
  static void replay(const BatchParams& P, bool materialize, hipStream_t s) {
    dim3 grid((P.nchunk + 63) / 64, P.B);
    if (materialize) {
      if (P.fast_trig)
        hipLaunchKernelGGL((replay_kernel<JR, JC, true, true>), grid, dim3(64), 0, s, P);
      else
        hipLaunchKernelGGL((replay_kernel<JR, JC, true, false>), grid, dim3(64), 0, s, P);
    } else {
      if (P.fast_trig)
        hipLaunchKernelGGL((replay_kernel<JR, JC, false, true>), grid, dim3(64), 0, s, P);
      else
        hipLaunchKernelGGL((replay_kernel<JR, JC, false, false>), grid, dim3(64), 0, s, P);
    }
  }
  static BatchLaunchers table() {
    return BatchLaunchers{&summarize, &prefix, &replay, Widths<JR, JC>::ELEM,
                          Widths<JR, JC>::START};
  }
};

// Per-problem reduction of the chunk partials + the -inf rules (api.hip).
void launch_finalize(const BatchParams& P, hipStream_t s);
// [problem][n] -> [problem][i][chunk] for n = chunk * L + i (api.hip).
void launch_relayout(const double* src, long src_stride, double* dst, long dst_stride, int nsrc,
                     int N, int L, int nchunk, hipStream_t s);

// Filled by the per-width translation units (batch_w*.hip).
const BatchLaunchers* find_batch_launchers(int JR, int JC);

}  // namespace clr
