// celerite_amd/csrc/clr_grad_kernels.h -- device side of the chunk-parallel gradient (clr_grad_core.h), widths 1..8.
// Included from the middle of clr_batch_kernels.h (BatchParams, make_direct, load_problem are in scope).
//
// After an evaluation by the scan (start state of every chunk in P.starts, route per problem in P.need_exact):
//   grad_riders_kernel    lane = chunk: AA, eta, JJ of the chunk along the base trajectory          [B][nchunk][RID]
//   grad_tangent_kernel   lane = chunk, blockIdx.z = direction group: the base recurrence + the group's two
//                         tangents from zero tangent states                                      [B][nchunk][NG][OUT]
//   grad_combine_kernel   thread = (problem, direction): walks the chunks                                 [B][NG]
// Reverse mode (the default, clr_batch_grad): grad_riders_kernel also stores w, D, x per sample, the state wherever the
// accumulated decay asks for one (GradStore, clr_grad_core.h) and the state after the chunk; then
//   grad_adjoint_kernel   thread = problem: the adjoint at every chunk end, backwards over the riders      [B][nchunk][ADJ]
//   grad_backward_kernel  lane = chunk: the reverse sweep, ALL partials at once                             [B][nchunk][NG]
//   grad_reduce_kernel    thread = (problem, direction): sums the chunks; per problem the largest drift of the
//                         reconstructed states (grad_backward_chunk's certificate)
// Problems the scan handed to the sequential recurrence (need_exact >= 2: their scanned start states are not
// certified) are skipped; the host runs the sequential gradient kernel (grad_kernels.hip) for them.  g_mask (may be
// null) restricts the forward-mode kernels to the problems whose reverse sweep drifted.
// The series is read through DirectSeries: from the chunk-interleaved copy when there is one (one 512-B line per array,
// step and wave), else from the row-major arrays (8-B reads that re-fetch their 64-B lines: 5x the bytes, measured, but
// at ~900 fp64 instructions per step and wave not what the forward-mode kernels wait for).
// A gradient chunk is P.g_m consecutive chunks of the scan (every scan chunk's start state is a valid start): the
// host balances the tangent pass (steps per lane) against the walk over the gradient chunks (clr_batch_grad) -- one
// long series is scanned in ~2000 chunks but differentiated in ~250.
#pragma once

#include "clr_grad_core.h"

namespace clr {

// gradient chunk c of problem b: on the chunk-interleaved copy of the series when the plan has one and a gradient
// chunk IS a scan chunk (the host sets P up that way: lane_cs == 1), else on the row-major arrays
__device__ __forceinline__ DirectSeries grad_series(const BatchParams& P, int b, int c) {
  if (P.lane_cs == 1) return make_direct(P, b, c);
  const long Lg = (long)P.g_m * P.L, first = (long)c * Lg;
  return DirectSeries{P.t + b * P.t_stride + first, P.diag + b * P.diag_stride + first, P.y + b * P.y_stride + first,
                      1, Lg, (int)Lg, (long)P.N - first};
}

// the stored states of gradient chunk c of problem b (reverse mode; null when nothing is recorded)
__device__ __forceinline__ GradStore grad_store(const BatchParams& P, int b, int c) {
  GradStore st;
  if (P.g_rec) {
    st.ck = P.g_ck + b * P.g_ck_stride + c;
    st.flag = P.g_ckflag + ((long)b * gridDim.x + blockIdx.x) * ((long)P.g_m * P.L);
    st.K = P.g_K;
    st.nalloc = P.g_nalloc;
    st.span = P.g_span > 1 ? P.g_span : 1;
    st.count = P.g_count + (long)b * P.g_nchunk + c;
  }
  return st;
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
                                  P.g_nchunk, P.g_rec ? P.g_ends + slot * Wd::START : nullptr, grad_store(P, b, c));
}

// Reverse mode when a gradient chunk is a scan chunk: the riders from the scan's own elements (one Gauss-Jordan per
// chunk, grad_riders_from_element) and the record by the plain recurrence (grad_riders_chunk without its riders).
template <int J>
__global__ void __launch_bounds__(64) grad_riders_elem_kernel(const BatchParams P) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  const int b = blockIdx.y;
  if (P.need_exact[b] >= 2) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const long slot = (long)b * P.nchunk + c;
  double* out = P.g_riders + slot * RID;
  grad_riders_from_element<J>(P.elems + slot * ELEM, c > 0 ? P.starts + slot * START : nullptr, out);
  if (c == P.nchunk - 1) {
    // the last chunk's A is never used by the scan and may hold anything behind the end of the series (padding);
    // its AA only ever multiplies the zero adjoint at the end of the series
#pragma unroll
    for (int i = 0; i < J * J; ++i) out[i] = 0.0;
  }
}

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) grad_record_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  const int b = blockIdx.y;
  if (P.need_exact[b] >= 2) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.g_nchunk) return;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  DirectSeries src = grad_series(P, b, c);
  const long slot = (long)b * P.g_nchunk + c;
  grad_riders_chunk<JR, JC, FAST, DirectSeries, false>(
      p, src, P.g_m * P.L, P.N, c * P.g_m * P.L,
      c > 0 ? P.starts + ((long)b * P.nchunk + (long)c * P.g_m) * Wd::START : nullptr, nullptr,
      P.g_rec + b * P.g_rec_stride + c, P.g_nchunk, P.g_ends + slot * Wd::START, grad_store(P, b, c));
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

// grad_adjoint_walk (clr_grad_core.h) with one WAVE per segment of a problem's chunks: lane (i, j) owns entry (i, j) of
// the J x J products, operands through LDS, the next chunk's riders prefetched into registers while the current chunk is
// multiplied.  One thread per problem walked a chunk in ~6.6 us (0.42 ms at 64 chunks, the longest phase of a single
// long series); a wave takes ~0.17 us.
// A segment is `seg` consecutive entries of `riders` ([B][nent][RID]); it starts from seg_start[b][k] (the adjoint at the
// END of its last entry; null: zeros -- the end of the series) and writes the adjoint at the end of each of its entries
// into adj ([B][nent][ADJ]).  One segment = the whole walk (rounds 3's kernel).  Round 4, one long series (thousands of
// gradient chunks, 0.70 ms of a 1.97 ms call as one walk): two levels -- the riders of a group of chunks compose to the
// riders of the merged chunk (grad_riders_compose_kernel), the groups are walked, and every group walks its own chunks
// from the adjoint the level above found for its end.
struct AdjointWalk {
  const double* riders;
  const double* seg_start;  // [B][nseg][ADJ] or null
  double* adj;
  const int* need_exact;
  int nent, seg;
};
template <int J>
__global__ void __launch_bounds__(64) grad_adjoint_kernel(const AdjointWalk W) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ, ADJ = SZ + J, JJN = J * J;
  __shared__ double AA[JJN], eta[J], JJ[SZ], Sb[JJN], fb[J], g[J], T[JJN];
  const int b = blockIdx.y, k = blockIdx.x, l = threadIdx.x;
  if (W.need_exact[b] >= 2) return;
  const int i = l / J, j = l % J;  // (lanes >= J * J idle)
  const bool mat = l < JJN;
  const int nseg = gridDim.x, lo = k * W.seg, hi = (lo + W.seg < W.nent) ? lo + W.seg : W.nent;
  const double* riders = W.riders + (long)b * W.nent * RID;
  double* adj = W.adj + (long)b * W.nent * ADJ;
  const double* st = W.seg_start ? W.seg_start + ((long)b * nseg + k) * ADJ : nullptr;
  if (mat) Sb[l] = st ? st[sym(i, j)] : 0.0;
  if (l < J) fb[l] = st ? st[SZ + l] : 0.0;
  int c = hi - 1;
  double r_aa = 0.0, r_eta = 0.0, r_jj = 0.0;
  auto fetch = [&](int cc) {
    const double* R = riders + (long)cc * RID;
    if (mat) r_aa = R[l];
    if (l < J) r_eta = R[JJN + l];
    if (l < SZ) r_jj = R[JJN + J + l];
  };
  if (c > lo) fetch(c);
  __syncthreads();
  for (; c >= lo; --c) {
    // the adjoint at the end of chunk c
    if (mat && i <= j) adj[(long)c * ADJ + tri(i, j)] = Sb[l];
    if (l < J) adj[(long)c * ADJ + SZ + l] = fb[l];
    if (c == lo) break;
    if (mat) AA[l] = r_aa;
    if (l < J) eta[l] = r_eta;
    if (l < SZ) JJ[l] = r_jj;
    if (c > lo + 1) fetch(c - 1);
    __syncthreads();
    if (l < J) {
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < J; ++q) a = fma(AA[q * J + l], fb[q], a);
      g[l] = a;
    }
    if (mat) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < J; ++q) t = fma(Sb[i * J + q], AA[q * J + j], t);
      T[l] = t;
    }
    __syncthreads();
    double s_new = 0.0, f_new = 0.0;
    if (mat) {
      s_new = eta[i] * eta[j] - JJ[sym(i, j)] - 0.5 * (g[i] * eta[j] + eta[i] * g[j]);
#pragma unroll
      for (int q = 0; q < J; ++q) s_new = fma(AA[q * J + i], T[q * J + j], s_new);
    }
    if (l < J) f_new = g[l] - 2.0 * eta[l];
    __syncthreads();
    if (mat) Sb[l] = s_new;
    if (l < J) fb[l] = f_new;
    __syncthreads();
  }
}

// The riders of a GROUP of consecutive chunks = the riders of the merged chunk (clr_grad_core.h, header: AA is the
// product of the steps' F, eta and JJ sum r x / D and r r^T / D with r pulled back through the steps before): for chunk a
// followed by chunk b
//     AA_ab = AA_b AA_a        eta_ab = eta_a + AA_a^T eta_b        JJ_ab = JJ_a + AA_a^T JJ_b AA_a .
// One wave per (group, problem), the group's chunks in order; group 0 is skipped (nothing lies before it: the walk over
// the groups never applies it).  out: [B][ngroup][RID].
template <int J>
__global__ void __launch_bounds__(64) grad_riders_compose_kernel(const double* riders_all, double* out, const int* need_exact,
                                                                 int nent, int seg) {
  constexpr int SZ = J * (J + 1) / 2, RID = J * J + J + SZ, JJN = J * J;
  __shared__ double A[JJN], Bm[JJN], eb[J], JB[JJN], T[JJN];
  const int b = blockIdx.y, k = blockIdx.x, l = threadIdx.x;
  if (k == 0 || need_exact[b] >= 2) return;
  const int i = l / J, j = l % J;
  const bool mat = l < JJN;
  const int lo = k * seg, hi = (lo + seg < nent) ? lo + seg : nent;
  const double* riders = riders_all + (long)b * nent * RID;
  // the running composition: lane (i, j) holds AA[i][j], JJ[i][j] (full, symmetric); lane i < J holds eta[i]
  double aa = 0.0, jj = 0.0, et = 0.0;
  {
    const double* R = riders + (long)lo * RID;
    if (mat) { aa = R[l]; jj = R[JJN + J + sym(i, j)]; }
    if (l < J) et = R[JJN + l];
  }
  for (int c = lo + 1; c < hi; ++c) {
    const double* R = riders + (long)c * RID;
    if (mat) { A[l] = aa; Bm[l] = R[l]; JB[l] = R[JJN + J + sym(i, j)]; }
    if (l < J) eb[l] = R[JJN + l];
    __syncthreads();
    double aa_new = 0.0, t = 0.0;
    if (mat) {
#pragma unroll
      for (int q = 0; q < J; ++q) {
        aa_new = fma(Bm[i * J + q], A[q * J + j], aa_new);  // AA_b AA_a
        t = fma(JB[i * J + q], A[q * J + j], t);             // JJ_b AA_a
      }
      T[l] = t;
    }
    if (l < J) {
#pragma unroll
      for (int q = 0; q < J; ++q) et = fma(A[q * J + l], eb[q], et);  // eta_a + AA_a^T eta_b
    }
    __syncthreads();
    if (mat) {
#pragma unroll
      for (int q = 0; q < J; ++q) jj = fma(A[q * J + i], T[q * J + j], jj);  // JJ_a + AA_a^T (JJ_b AA_a)
      aa = aa_new;
    }
    __syncthreads();
  }
  double* O = out + ((long)b * gridDim.x + k) * RID;
  if (mat) {
    O[l] = aa;
    if (i <= j) O[JJN + J + tri(i, j)] = jj;
  }
  if (l < J) O[JJN + l] = et;
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
                                    P.g_part + slot * Sh::NG, P.g_adj0 + slot * Wd::START, grad_store(P, b, c), &drift,
                                    c > 0 ? P.starts + ((long)b * P.nchunk + (long)c * P.g_m) * Wd::START : nullptr);
  if (P.g_K == 1) drift = 0.0;  // (every state is a stored one: no reconstructed state is used)
  P.g_drift[slot] = drift;
}

// The partials of a problem = -1/2 of the sum over its gradient chunks, and the two certificates of the reverse sweep
// reduced to one number per problem --
//   * the drift of the reconstructed states (g_drift, zero when every state is stored), and
//   * the adjoint a sweep arrived at for its chunk's first sample (g_adj0) against the one the walk over the riders
//     predicted for the end of the previous chunk (g_adj): two independent computations of one quantity -- it is what
//     vouches for riders taken from the scan's elements;
// a non-finite partial or a NaN anywhere is a failed certificate (reported as +inf).
// Two kernels (round 4: one workgroup per problem took 0.27-0.32 ms for the 4224 chunks of one long series):
//   grad_reduce_slab_kernel   workgroup = (slab of 256 chunks, problem).  Thread (stripe s = tid / 32, q = tid % 32 < NG)
//                             sums partial q over the slab's chunks c = s, s + 8, ... in order, the eight stripes are
//                             added in a fixed tree; thread = chunk: that chunk's certificate, reduced to the slab's
//                             maximum.                                              -> g_slab [B][nslab][33]
//   grad_reduce_kernel        one wave per problem: the slabs in order.
// The shape of the sums depends on the chunk count only -- not on the batch size or the sharding -- so results stay
// bit-identical across shard counts.
template <int J>
__global__ void __launch_bounds__(256) grad_reduce_slab_kernel(const BatchParams P, int NG) {
  constexpr int ADJ = J * (J + 1) / 2 + J;
  __shared__ double stripe[8][32], wmax[4];
  const int b = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
  if (P.need_exact[b] >= 2) return;  // (workgroup-uniform)
  const int ng = P.g_nchunk, c0 = slab * 256, c1 = (c0 + 256 < ng) ? c0 + 256 : ng;
  {
    const int q = tid & 31, s = tid >> 5;
    double acc = 0.0;
    if (q < NG) {
      const double* part = P.g_part + (long)b * ng * NG + q;
      for (int c = c0 + s; c < c1; c += 8) acc += part[(long)c * NG];
    }
    stripe[s][q] = acc;
  }
  double worst = 0.0;
  bool bad = false;
  const int c = c0 + tid;
  if (c < c1) {
    const long slot = (long)b * ng + c;
    const double d = P.g_drift[slot];
    if (d != d) bad = true;
    worst = fmax(worst, d);
    for (int k = 0; k < NG; ++k)
      if (!(fabs(P.g_part[slot * NG + k]) <= 1.79e308)) bad = true;
    if (c > 0) {
      const double *want = P.g_adj + (slot - 1) * ADJ, *got = P.g_adj0 + slot * ADJ;
      double big = 0.0, dev = 0.0;
#pragma unroll
      for (int i = 0; i < ADJ; ++i) {
        const double a = fabs(want[i]), e = fabs(want[i] - got[i]);
        if (a != a || e != e) bad = true;
        big = fmax(big, a);
        dev = fmax(dev, e);
      }
      worst = fmax(worst, big > 0.0 ? dev / big : (dev == 0.0 ? 0.0 : INFINITY));
    }
  }
  if (bad) worst = INFINITY;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
  if ((tid & 63) == 0) wmax[tid >> 6] = worst;
  __syncthreads();
  double* o = P.g_slab + ((long)b * gridDim.x + slab) * 33;
  if (tid < 32) {
    o[tid] = ((stripe[0][tid] + stripe[1][tid]) + (stripe[2][tid] + stripe[3][tid])) +
             ((stripe[4][tid] + stripe[5][tid]) + (stripe[6][tid] + stripe[7][tid]));
  }
  if (tid == 0) o[32] = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
}

template <int J>
__global__ void __launch_bounds__(64) grad_reduce_kernel(const BatchParams P, int NG, int nslab) {
  const int b = blockIdx.x, l = threadIdx.x;
  if (P.need_exact[b] >= 2) return;
  const double* S = P.g_slab + (long)b * nslab * 33;
  if (l < NG) {
    double acc = 0.0;
    for (int k = 0; k < nslab; ++k) acc += S[k * 33 + l];
    P.g_res[(long)b * NG + l] = -0.5 * acc;
  }
  double worst = 0.0;
  for (int k = l; k < nslab; k += 64) worst = fmax(worst, S[k * 33 + 32]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
  if (l == 0) P.g_drift_max[b] = worst;
}

}  // namespace clr
