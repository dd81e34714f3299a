// celerite_amd/csrc/wide_grad_kernels.hip -- grad_log_likelihood PARALLEL IN n at widths 9..64 and with general terms
// (the reference: celerite/solver.cpp:347-463, forward-mode AD of cholesky.h:41-210 + :326-401 at any width).
//
// DESIGN.md section 3.1 splits a tangent recurrence over a chunk that starts from the TRUE base state into the
// tangent from a ZERO tangent state (independent per chunk and direction: wide_grad_kernel<.., CHUNKED>,
// grad_kernels.hip) and the homogeneous propagation of the tangent state the chunk starts with, through three riders
// of the base trajectory shared by all directions:
//     dS_end = AA dS0 AA^T + dS_end0          df_end = AA (df0 - dS0 eta) + df_end0
//     d(log det) = d(log det)0 - <JJ, dS0>    d(quad) = d(quad)0 - 2 eta.df0 + eta^T dS0 eta
// Widths 1..8 keep those objects in one lane's registers (clr_grad_core.h).  At the padded widths JP = 16 / 32 (round 6:
// 64, wide_grad_riders64_kernel below) they are JP x JP matrices in LDS and the work is a wave's:
//   wide_grad_riders_kernel   one wave per (problem, chunk): the riders from the wide scan's OWN element of the chunk
//                             (A, eta_e, Jm from the zero state) and the chunk's scanned start state (P, f) -- they are
//                             the derivatives of the element's maps at that state (grad_riders_from_element):
//                             Mi = (I + P Jm)^-1 by Gauss-Jordan with partial pivoting on [I + P Jm | I] (lane = column),
//                             AA = A Mi, eta = Mi^T (Jm f - eta_e), JJ = -sym(Jm Mi);
//   wide_grad_walk_kernel     one wave per (problem, direction): the chunks in order, two JP^3 products per chunk.
// Problems the evaluation routed to the sequential recurrence (their scanned start states are not certified) are left
// to the sequential tangent kernel (launch_grad, only_level).
#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_wide.h"

namespace clr {

namespace {

__device__ __forceinline__ void wg_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }  // (one wave per workgroup)

__device__ __forceinline__ double wg_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int JP>
__global__ void __launch_bounds__(64) wide_grad_riders_kernel(const WideGradWalk W) {
  constexpr int SZ = JP * (JP + 1) / 2, ELEM = JP * JP + JP + SZ + JP + SZ, START = SZ + JP;
  constexpr int LD = JP + 1, LT = 2 * JP + 2, RID = 2 * JP * JP + JP;
  __shared__ double Pm[JP * LD], Jf[JP * LD], Am[JP * LD], X[JP * LD], T[JP * LT], f0[JP], ee[JP], wv[JP];
  const int lane = threadIdx.x;
  const long slot = blockIdx.x;
  const int b = (int)(slot / W.nchunk), c = (int)(slot % W.nchunk);
  if (W.level && W.level[b] >= 2) return;
  const double* E = W.elems + slot * ELEM;
  const double* Eeta = E + JP * JP + JP + SZ;
  const double* EJm = Eeta + JP;
  const double* st = W.starts + slot * START;
  for (int idx = lane; idx < JP * JP; idx += 64) {
    const int i = idx / JP, j = idx % JP;
    Pm[i * LD + j] = c > 0 ? st[sym(i, j)] : 0.0;   // (the first chunk starts from the zero state)
    Jf[i * LD + j] = EJm[sym(i, j)];
    Am[i * LD + j] = E[idx];
  }
  if (lane < JP) { f0[lane] = c > 0 ? st[SZ + lane] : 0.0; ee[lane] = Eeta[lane]; }
  wg_fence();
  // [ I + P Jm | I ]
  for (int idx = lane; idx < JP * JP; idx += 64) {
    const int i = idx / JP, j = idx % JP;
    double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll 8
    for (int k = 0; k < JP; ++k) acc = fma(Pm[i * LD + k], Jf[k * LD + j], acc);
    T[i * LT + j] = acc;
    T[i * LT + JP + j] = (i == j) ? 1.0 : 0.0;
  }
  if (lane < JP) {  // wv = Jm f - eta_e
    double acc = -ee[lane];
#pragma unroll 8
    for (int k = 0; k < JP; ++k) acc = fma(Jf[lane * LD + k], f0[k], acc);
    wv[lane] = acc;
  }
  wg_fence();
  // Gauss-Jordan with partial pivoting, lane = column of the tableau (2 JP <= 64 columns)
  const bool col_lane = lane < 2 * JP;
  for (int col = 0; col < JP; ++col) {
    int piv = col;
    double best = -1.0;
    for (int i = col; i < JP; ++i) {  // (every lane scans the pivot column: broadcast reads)
      const double cand = fabs(T[i * LT + col]);
      if (cand > best) { best = cand; piv = i; }
    }
    const double top = col_lane ? T[piv * LT + lane] : 0.0, old = col_lane ? T[col * LT + lane] : 0.0;
    const double pv = T[piv * LT + col];
    wg_fence();
    const double t = top * (1.0 / pv);
    if (col_lane) T[piv * LT + lane] = old;  // row swap (a no-op when piv == col) ...
    wg_fence();
    if (col_lane) T[col * LT + lane] = t;    // ... and the scaled pivot row
    wg_fence();
    for (int i = 0; i < JP; ++i) {
      if (i == col) continue;
      const double m = T[i * LT + col];      // (broadcast; this lane's own column is updated below)
      wg_fence();
      if (col_lane) T[i * LT + lane] = fma(-m, t, T[i * LT + lane]);
    }
    wg_fence();
  }
  // Mi = T[:, JP:]:  AA = A Mi ;  X = Jm Mi ;  eta = Mi^T wv
  double* out = W.riders + slot * RID;
  for (int idx = lane; idx < JP * JP; idx += 64) {
    const int i = idx / JP, j = idx % JP;
    double a = 0.0, x = 0.0;
#pragma unroll 8
    for (int k = 0; k < JP; ++k) {
      const double mi = T[k * LT + JP + j];
      a = fma(Am[i * LD + k], mi, a);
      x = fma(Jf[i * LD + k], mi, x);
    }
    out[idx] = a;
    X[i * LD + j] = x;
  }
  if (lane < JP) {
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < JP; ++k) acc = fma(T[k * LT + JP + lane], wv[k], acc);
    out[JP * JP + lane] = acc;
  }
  wg_fence();
  for (int idx = lane; idx < JP * JP; idx += 64) {  // JJ = -sym(Jm Mi)
    const int i = idx / JP, j = idx % JP;
    out[JP * JP + JP + idx] = -0.5 * (X[i * LD + j] + X[j * LD + i]);
  }
}

// The riders at the padded width 64 (round 6): the tableau [I + P Jm | I] alone is 66 KB, so a workgroup of 256 threads per
// (problem, chunk) eliminates it in 133 KB of dynamic LDS -- P's buffer is reused for A once the tableau is formed, the
// tableau's left half for X = Jm Mi once the elimination is done (the arrangement of wide_walk_kernel<64>, wide64_kernels.hip).
__global__ void __launch_bounds__(256) wide_grad_riders64_kernel(const WideGradWalk W) {
  constexpr int J = 64, SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  constexpr int LD = J + 1, LT = 2 * J + 2, RID = 2 * J * J + J, NT = 256;
  extern __shared__ double rid_lds[];
  double* Pm = rid_lds;          // [J][LD] the start state P; later A
  double* Jf = Pm + J * LD;      // [J][LD] Jm
  double* T = Jf + J * LD;       // [J][LT] [ I + P Jm | I ] -> [ . | Mi ]; the left half later X = Jm Mi
  double* f0 = T + J * LT;       // [J]
  double* wv = f0 + J;           // [J] Jm f - eta_e
  const int tid = threadIdx.x, lane = tid & 63;
  const long slot = blockIdx.x;
  const int b = (int)(slot / W.nchunk), c = (int)(slot % W.nchunk);
  if (W.level && W.level[b] >= 2) return;
  if (c == 0) {  // the first chunk starts from the zero state with a zero tangent: its riders only ever multiply zeros (and
                 // the riderless first chunk of the wide summarize does not form A, eta, Jm at all)
    double* o = W.riders + slot * RID;
    for (int idx = tid; idx < RID; idx += NT) o[idx] = 0.0;
    return;
  }
  const double* E = W.elems + slot * ELEM;
  const double* Eeta = E + J * J + J + SZ;
  const double* EJm = Eeta + J;
  const double* st = W.starts + slot * START;
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    Pm[i * LD + j] = c > 0 ? st[sym(i, j)] : 0.0;   // (the first chunk starts from the zero state)
    Jf[i * LD + j] = EJm[sym(i, j)];
  }
  if (tid < J) f0[tid] = c > 0 ? st[SZ + tid] : 0.0;
  __syncthreads();
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(Pm[i * LD + k], Jf[k * LD + j], acc);
    T[i * LT + j] = acc;
    T[i * LT + J + j] = (i == j) ? 1.0 : 0.0;
  }
  if (tid < J) {
    double acc = -Eeta[tid];
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(Jf[tid * LD + k], f0[k], acc);
    wv[tid] = acc;
  }
  __syncthreads();
  // Gauss-Jordan with partial pivoting: thread = (column cc of the tableau, half of the rows)
  const int cc = tid & 127, rg = tid >> 7;
  for (int col = 0; col < J; ++col) {
    double best = lane >= col ? fabs(T[lane * LT + col]) : -1.0;  // (every wave searches: the first one on ties)
    int piv = lane;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const double ob = __shfl_xor(best, m, 64);
      const int op = __shfl_xor(piv, m, 64);
      const bool take = ob > best || (ob == best && op < piv);
      best = take ? ob : best;
      piv = take ? op : piv;
    }
    const double inv = 1.0 / T[piv * LT + col];
    const double top = T[piv * LT + cc], old = T[col * LT + cc];
    const double t = top * inv;
    __syncthreads();  // (everyone has read rows piv and col)
    if (rg == 0) {
      T[piv * LT + cc] = old;                   // the row swap (a no-op when piv == col) ...
      T[col * LT + cc] = (cc > col) ? t : top;  // ... and the scaled pivot row (columns <= col are never read again)
    }
    __syncthreads();
    if (cc > col) {
      for (int i = rg; i < J; i += 2)
        if (i != col) T[i * LT + cc] = fma(-T[i * LT + col], t, T[i * LT + cc]);
    }
    __syncthreads();
  }
  // Mi = T[:, J:]:  AA = A Mi ;  X = Jm Mi ;  eta = Mi^T wv
  for (int idx = tid; idx < J * J; idx += NT) Pm[(idx / J) * LD + idx % J] = E[idx];  // (A; P is no longer needed)
  __syncthreads();
  double* out = W.riders + slot * RID;
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    double a = 0.0, x = 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) {
      const double mi = T[k * LT + J + j];
      a = fma(Pm[i * LD + k], mi, a);
      x = fma(Jf[i * LD + k], mi, x);
    }
    out[idx] = a;
    T[i * LT + j] = x;  // (the left half of the tableau is free)
  }
  if (tid < J) {
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(T[k * LT + J + tid], wv[k], acc);
    out[J * J + tid] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < J * J; idx += NT) {  // JJ = -sym(Jm Mi)
    const int i = idx / J, j = idx % J;
    out[J * J + J + idx] = -0.5 * (T[i * LT + j] + T[j * LT + i]);
  }
}

template <int JP>
__global__ void __launch_bounds__(64) wide_grad_walk_kernel(const WideGradWalk W) {
  constexpr int LD = JP + 1, RID = 2 * JP * JP + JP, OUT = JP * JP + JP + 2;
  extern __shared__ double walk_lds[];  // (dynamic: 3 x 33 KB at JP = 64)
  double* dS = walk_lds;
  double* Tm = dS + JP * LD;
  double* AA = Tm + JP * LD;
  double* df = AA + JP * LD;
  double* h = df + JP;
  double* tmp = h + JP;
  double* eta = tmp + JP;
  const int lane = threadIdx.x, p = blockIdx.x, b = blockIdx.y;
  if (W.level && W.level[b] >= 2) return;  // (the sequential tangent kernel writes this problem's results)
  if (W.ll_status[b] != CLR_OK) {          // quiet semantics: -inf, zero gradient (celerite.py:205-208)
    if (lane == 0) {
      W.out_grad[(long)b * W.NG + p] = 0.0;
      if (p == 0) { W.out_value[b] = -INFINITY; W.out_status[b] = W.ll_status[b]; }
    }
    return;
  }
  for (int idx = lane; idx < JP * LD; idx += 64) dS[idx] = 0.0;
  if (lane < JP) df[lane] = 0.0;
  wg_fence();
  double dld = 0.0, dqd = 0.0;
  for (int c = 0; c < W.nchunk; ++c) {
    const double* R = W.riders + ((long)b * W.nchunk + c) * RID;
    const double* G = W.rec + (((long)b * W.nchunk + c) * W.NG + p) * OUT;
    double acc = 0.0;
    for (int idx = lane; idx < JP * JP; idx += 64) {
      const int i = idx / JP, j = idx % JP;
      AA[i * LD + j] = R[idx];
      acc = fma(R[JP * JP + JP + idx], dS[i * LD + j], acc);  // <JJ, dS> over ALL entries
    }
    if (lane < JP) eta[lane] = R[JP * JP + lane];
    wg_fence();
    double t = 0.0, e1 = 0.0, e2 = 0.0;
    if (lane < JP) {
#pragma unroll 8
      for (int k = 0; k < JP; ++k) t = fma(dS[lane * LD + k], eta[k], t);
      tmp[lane] = t;
      e1 = eta[lane] * df[lane];
      e2 = eta[lane] * t;
    }
    acc = wg_sum(acc); e1 = wg_sum(e1); e2 = wg_sum(e2);
    dld += G[JP * JP + JP] - acc;
    dqd += G[JP * JP + JP + 1] - 2.0 * e1 + e2;
    if (c + 1 == W.nchunk) break;
    if (lane < JP) h[lane] = df[lane] - tmp[lane];
    wg_fence();
    if (lane < JP) {  // df' = df_end0 + AA (df - dS eta)
      double a = G[JP * JP + lane];
#pragma unroll 8
      for (int k = 0; k < JP; ++k) a = fma(AA[lane * LD + k], h[k], a);
      df[lane] = a;
    }
    for (int idx = lane; idx < JP * JP; idx += 64) {  // Tm = AA dS
      const int i = idx / JP, j = idx % JP;
      double a = 0.0;
#pragma unroll 8
      for (int k = 0; k < JP; ++k) a = fma(AA[i * LD + k], dS[k * LD + j], a);
      Tm[i * LD + j] = a;
    }
    wg_fence();
    for (int idx = lane; idx < JP * JP; idx += 64) {  // dS' = dS_end0 + Tm AA^T
      const int i = idx / JP, j = idx % JP;
      double a = G[idx];
#pragma unroll 8
      for (int k = 0; k < JP; ++k) a = fma(Tm[i * LD + k], AA[j * LD + k], a);
      dS[i * LD + j] = a;
    }
    wg_fence();
  }
  if (lane == 0) {
    double g = -0.5 * (dqd + dld);
    if (p == 0 && !(W.jitter[b] > 2.220446049250313e-16)) g = 0.0;  // solver.cpp:379-389,419-426
    W.out_grad[(long)b * W.NG + p] = g;
    if (p == 0) {
      // the evaluation's log-likelihood with the reference's constant: -(quad + log det + pi log N) / 2 (solver.cpp:415)
      W.out_value[b] = W.ll[b] + 0.5 * W.N * 1.8378770664093453 - 0.5 * 3.14159265358979323846 * log((double)W.N);
      W.out_status[b] = CLR_OK;
    }
  }
}

}  // namespace

int launch_wide_grad_riders(const WideGradWalk& W, hipStream_t s) {
  const dim3 grid((unsigned)((long)W.B * W.nchunk));
  if (W.JP == 16) hipLaunchKernelGGL((wide_grad_riders_kernel<16>), grid, dim3(64), 0, s, W);
  else if (W.JP == 32) hipLaunchKernelGGL((wide_grad_riders_kernel<32>), grid, dim3(64), 0, s, W);
  else {
    const int bytes = (2 * 64 * 65 + 64 * 130 + 2 * 64) * (int)sizeof(double);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_grad_riders64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return 1;
    hipLaunchKernelGGL(wide_grad_riders64_kernel, grid, dim3(256), bytes, s, W);
  }
  return 0;
}

int launch_wide_grad_walk(const WideGradWalk& W, hipStream_t s) {
  const dim3 grid(W.NG, W.B);
  const int bytes = (3 * W.JP * (W.JP + 1) + 4 * W.JP) * (int)sizeof(double);
  if (W.JP == 16) hipLaunchKernelGGL((wide_grad_walk_kernel<16>), grid, dim3(64), bytes, s, W);
  else if (W.JP == 32) hipLaunchKernelGGL((wide_grad_walk_kernel<32>), grid, dim3(64), bytes, s, W);
  else {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_grad_walk_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return 1;
    hipLaunchKernelGGL((wide_grad_walk_kernel<64>), grid, dim3(64), bytes, s, W);
  }
  return 0;
}

}  // namespace clr
