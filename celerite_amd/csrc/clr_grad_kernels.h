// celerite_amd/csrc/clr_grad_kernels.h -- device side of the chunk-parallel gradient (clr_grad_core.h), widths 1..8.
// Included from the middle of clr_batch_kernels.h (BatchParams, make_direct, load_problem are in scope).
//
// After an evaluation by the scan (start state of every chunk in P.starts, route per problem in P.need_exact):
//   grad_riders_kernel    lane = chunk: AA, eta, JJ of the chunk along the base trajectory          [B][nchunk][RID]
//   grad_tangent_kernel   lane = chunk, blockIdx.z = direction group: the base recurrence + the group's two
//                         tangents from zero tangent states                                      [B][nchunk][NG][OUT]
//   grad_combine_kernel   thread = (problem, direction): walks the chunks                                 [B][NG]
// Reverse mode (the default, clr_batch_grad): grad_riders_kernel also stores w, D, x per sample, the state every g_K
// steps and the state after the chunk; then
//   grad_adjoint_kernel   thread = problem: the adjoint at every chunk end, backwards over the riders      [B][nchunk][ADJ]
//   grad_backward_kernel  lane = chunk: the reverse sweep, ALL partials at once                             [B][nchunk][NG]
//   grad_reduce_kernel    thread = (problem, direction): sums the chunks; per problem the largest drift of the
//                         reconstructed states (grad_backward_chunk's certificate)
// Problems the scan handed to the sequential recurrence (need_exact >= 2: their scanned start states are not
// certified) are skipped; the host runs the sequential gradient kernel (grad_kernels.hip) for them.  g_mask (may be
// null) restricts the forward-mode kernels to the problems whose reverse sweep drifted.
// The series is read through DirectSeries on the row-major arrays: at ~900 fp64 instructions per step and wave the
// 24 bytes per lane and step are not what the kernel waits for.
// A gradient chunk is P.g_m consecutive chunks of the scan (every scan chunk's start state is a valid start): the
// host balances the tangent pass (steps per lane) against the walk over the gradient chunks (clr_batch_grad) -- one
// long series is scanned in ~2000 chunks but differentiated in ~250.
#pragma once

#include "clr_grad_core.h"

namespace clr {

// gradient chunk c of problem b on the row-major arrays
__device__ __forceinline__ DirectSeries grad_series(const BatchParams& P, int b, int c) {
  const long Lg = (long)P.g_m * P.L, first = (long)c * Lg;
  return DirectSeries{P.t + b * P.t_stride + first, P.diag + b * P.diag_stride + first, P.y + b * P.y_stride + first,
                      1, Lg, (int)Lg, (long)P.N - first};
}

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) grad_riders_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  const int b = blockIdx.y;
  if (P.need_exact[b] >= 2) return;  // (wave-uniform)
  if (P.g_mask && !P.g_mask[b]) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.g_nchunk) return;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  DirectSeries src = grad_series(P, b, c);
  const long slot = (long)b * P.g_nchunk + c;
  grad_riders_chunk<JR, JC, FAST>(p, src, P.g_m * P.L, P.N, c * P.g_m * P.L,
                                  c > 0 ? P.starts + ((long)b * P.nchunk + (long)c * P.g_m) * Wd::START : nullptr,
                                  P.g_riders + slot * Sh::RID, P.g_rec ? P.g_rec + b * P.g_rec_stride + c : nullptr,
                                  P.g_nchunk, P.g_rec ? P.g_ends + slot * Wd::START : nullptr,
                                  P.g_rec ? P.g_ck + b * P.g_ck_stride + c : nullptr, P.g_K);
}

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) grad_tangent_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  const int b = blockIdx.y;
  if (P.need_exact[b] >= 2) return;
  if (P.g_mask && !P.g_mask[b]) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.g_nchunk) return;
  const int group = blockIdx.z;
  int kind, term, q0, q1;
  grad_group<JR, JC>(group, &kind, &term, &q0, &q1);
  DirectSeries src = grad_series(P, b, c);
  const long slot = (long)b * P.g_nchunk + c;
  double* rec = P.g_out + slot * Sh::NG * Sh::OUT;
  grad_chunk<JR, JC, FAST>(P.a_real + (long)b * JR, P.c_real + (long)b * JR, P.a_comp + (long)b * JC,
                           P.b_comp + (long)b * JC, P.c_comp + (long)b * JC, P.d_comp + (long)b * JC, P.jitter[b], src,
                           P.g_m * P.L, P.N, c * P.g_m * P.L,
                           c > 0 ? P.starts + ((long)b * P.nchunk + (long)c * P.g_m) * Wd::START : nullptr, group,
                           rec + (long)q0 * Sh::OUT, q1 >= 0 ? rec + (long)q1 * Sh::OUT : nullptr);
}

template <int J>
__global__ void __launch_bounds__(64) grad_combine_kernel(const BatchParams P, int NG) {
  constexpr int SZ = J * (J + 1) / 2, OUT = SZ + J + 2, RID = J * J + J + SZ;
  const long idx = (long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long)P.B * NG) return;
  const int b = (int)(idx / NG), q = (int)(idx % NG);
  if (P.need_exact[b] >= 2) return;
  if (P.g_mask && !P.g_mask[b]) return;
  double dld, dq;
  grad_combine<J>(P.g_nchunk, P.g_riders + (long)b * P.g_nchunk * RID,
                  P.g_out + ((long)b * P.g_nchunk * NG + q) * OUT, (long)NG * OUT, &dld, &dq);
  P.g_res[idx] = -0.5 * (dq + dld);
}

template <int J>
__global__ void __launch_bounds__(64) grad_adjoint_kernel(const BatchParams P) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ, ADJ = SZ + J;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B || P.need_exact[b] >= 2) return;
  grad_adjoint_walk<J>(P.g_nchunk, P.g_riders + (long)b * P.g_nchunk * RID, P.g_adj + (long)b * P.g_nchunk * ADJ);
}

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) grad_backward_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  using Sh = GradShape<JR, JC>;
  const int b = blockIdx.y;
  if (P.need_exact[b] >= 2) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.g_nchunk) return;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  DirectSeries src = grad_series(P, b, c);
  const long slot = (long)b * P.g_nchunk + c;
  double drift = 0.0;
  grad_backward_chunk<JR, JC, FAST>(p, src, P.g_m * P.L, P.N, c * P.g_m * P.L, P.g_ends + slot * Wd::START,
                                    P.g_adj + slot * Wd::START, P.g_rec + b * P.g_rec_stride + c, P.g_nchunk,
                                    P.g_part + slot * Sh::NG, nullptr, P.g_ck + b * P.g_ck_stride + c, P.g_K, &drift,
                                    c > 0 ? P.starts + ((long)b * P.nchunk + (long)c * P.g_m) * Wd::START : nullptr);
  P.g_drift[slot] = P.g_K > 1 ? drift : 0.0;  // (K = 1: every state is a stored one, no reconstructed state is used)
}

// thread = (problem, direction): -1/2 of the sum over the chunks; direction 0 also reduces the drift
// (a template only so that every width's translation unit owns its instantiation)
template <int J>
__global__ void __launch_bounds__(64) grad_reduce_kernel(const BatchParams P, int NG) {
  const long idx = (long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long)P.B * NG) return;
  const int b = (int)(idx / NG), q = (int)(idx % NG);
  if (P.need_exact[b] >= 2) return;
  const double* part = P.g_part + (long)b * P.g_nchunk * NG + q;
  double acc = 0.0;
  for (int c = 0; c < P.g_nchunk; ++c) acc += part[(long)c * NG];
  P.g_res[idx] = -0.5 * acc;
  if (q == 0) {
    double worst = 0.0;
    for (int c = 0; c < P.g_nchunk; ++c) {
      const double d = P.g_drift[(long)b * P.g_nchunk + c];
      if (!(d <= worst)) worst = d;
    }
    P.g_drift_max[b] = worst;
  }
}

}  // namespace clr
