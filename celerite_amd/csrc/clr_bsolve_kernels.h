// celerite_amd/csrc/clr_bsolve_kernels.h -- K^-1 b for every problem of a plan from its materialised factor: the batched
// form of CholeskySolver::solve (cholesky.h:218-318), parallel in n (round 5).
//
// The reference runs, per right-hand side, a forward substitution, a division by D and a backward substitution
// (cholesky.h:240-259; phi_, u_, W_, D_ in its storage: u_(:, n-1) = U~(t_n), phi_(:, n) = the decay n -> n+1):
//     forward   f <- phi_{n-1} (f + W_{n-1} x_{n-1}) ;  x_n = b_n - u_{n-1} . f            n = 1 .. N-1
//     x <- x / D
//     backward  f <- phi_n (f + u_n x_{n+1}) ;  x_n -= W_n . f                              n = N-2 .. 0
// In terms of the factor's own SLOTS (slot n: phi[n] = decay n -> n+1, u[n] = U~(t_n), W[n], D[n] -- the plan's
// chunk-interleaved layout keeps them that way) both sweeps only touch one slot per sample:
//     forward   x_n = b_n - u[n] . g ;  g <- phi[n] (g + W[n] x_n)                          g = the f the NEXT sample reads
//     backward  h = phi[n] k ;  x_n -= W[n] . h ;  k <- h + u[n] x_n                        k = what the PREVIOUS sample reads
// and both are AFFINE in their state with transition matrices F_n = Phi_n (I - W_n u_n^T) (forward) and F_n^T (backward).
// So the time axis is cut into the plan's chunks and each sweep is a chunked affine scan:
//   1. bsolve_summarize   per chunk, the product M = F_{hi-1} ... F_lo (3 J^2 flops per step, ONCE: shared by every
//                         right-hand side and by both sweeps -- the backward chunk map is M^T) and the forward offset a;
//   2. bsolve_prefix      per (problem, right-hand side): the state every chunk starts from, a walk over the chunks;
//   3. bsolve_forward     the forward recurrence per chunk from its start state: x / D in place;
//   4. bsolve_back_offset the backward offset of every chunk (from k = 0: 3 J flops per step), prefix with M^T,
//   5. bsolve_backward    the backward recurrence per chunk: x.
// Lane = (problem, chunk) as in the scan kernels; the right-hand sides live in the chunk-interleaved layout of the series
// ([problem][rhs][i][chunk]: a wave's access is 512 contiguous bytes).  LEAN: the factor holds W and D only and phi, u are
// regenerated per step from the times and the coefficients (cholesky.h:127-147) -- 9 instead of 25 doubles per sample at
// width 8 through HBM, four times per solve.
// Conventions at the end of the series: sample N-1 has no successor -- its forward transition is defined as 0 (the
// state it would produce is never read) and so is its backward one (the reference starts the backward sweep with f = 0);
// padded samples n >= N are skipped.  M^T of the last chunk is then 0 as well, which is what the backward prefix needs.
#pragma once

namespace clr {

struct BSolveParams {
  int nrhs, r;          // right-hand sides of the call, the one this launch works on
  int lean;             // the factor holds W, D only
  int have_M;           // the chunk maps of this factor are already in M (an earlier solve formed them)
  double* xT;           // [B][nrhs][L][nchunk] right-hand sides in, solutions out
  double* M;            // [B][nchunk][J*J]
  double* off;          // [B][nrhs][nchunk][J] chunk offsets (forward, then backward)
  double* starts;       // [B][nrhs][nchunk][J] chunk start states (forward, then backward)
};

// one sample's slot of the factor for lane (problem b, chunk c)
template <int JR, int JC, bool LEAN, bool FAST>
struct FactorSlots {
  static constexpr int J = JR + 2 * JC;
  const double *phi, *u, *W, *D;  // the lane's column of the chunk-interleaved arrays
  long fstride;                   // = nchunk
  DirectSeries ts;                // (LEAN) the times
  Problem<JR, JC> p;
  __device__ __forceinline__ void get(int i, double* ph, double* uu, double* ww, double* d) const {
    *d = D[(long)i * fstride];
#pragma unroll
    for (int j = 0; j < J; ++j) ww[j] = W[((long)i * J + j) * fstride];
    if (LEAN) {
      const double tn = ts.t(i);
      double vv[J], phid[nz(JR + JC)];
      features_uv<JR, JC, FAST>(p, tn, uu, vv);
      features_phi_distinct<JR, JC>(p, ts.t(i + 1) - tn, phid);
#pragma unroll
      for (int j = 0; j < J; ++j) ph[j] = phid[phi_index<JR>(j)];
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        ph[j] = phi[((long)i * J + j) * fstride];
        uu[j] = u[((long)i * J + j) * fstride];
      }
    }
  }
};

template <int JR, int JC, bool LEAN, bool FAST>
__device__ __forceinline__ FactorSlots<JR, JC, LEAN, FAST> make_slots(const BatchParams& P, int b, int c) {
  constexpr int J = JR + 2 * JC;
  FactorSlots<JR, JC, LEAN, FAST> s;
  const long cells = (long)P.L * P.nchunk;
  s.fstride = P.nchunk;
  s.W = P.W + (long)b * J * cells + c;
  s.D = P.D + (long)b * cells + c;
  s.phi = LEAN ? nullptr : P.phi + (long)b * J * cells + c;
  s.u = LEAN ? nullptr : P.u + (long)b * J * cells + c;
  if (LEAN) {
    s.ts = DirectSeries{P.t + b * P.t_stride + c * P.lane_cs, nullptr, nullptr, P.lane_is, P.lane_cs, P.L, (long)P.N - (long)c * P.L};
    load_problem<JR, JC>(P, b, s.p);
  }
  return s;
}

// 1. chunk maps M (WITH_M) and forward offsets a of right-hand side S.r
template <int JR, int JC, bool LEAN, bool FAST, bool WITH_M>
__global__ void __launch_bounds__(64) bsolve_summarize_kernel(const BatchParams P, const BSolveParams S) {
  constexpr int J = JR + 2 * JC;
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const auto F = make_slots<JR, JC, LEAN, FAST>(P, b, c);
  const long cells = (long)P.L * P.nchunk;
  const double* x = S.xT + ((long)b * S.nrhs + S.r) * cells + c;
  double M[WITH_M ? J * J : 1], a[J];
  if (WITH_M) {
#pragma unroll
    for (int i = 0; i < J * J; ++i) M[i] = (i / J == i % J) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int j = 0; j < J; ++j) a[j] = 0.0;
  const int n0 = c * P.L;
  // (the slot of step i + 1 and its right-hand side are fetched while step i is computed: a lone wave per SIMD has
  //  nothing else to hide the ~0.5 us of an HBM round trip behind)
  double nph[J], nuu[J], nww[J], nd, nb;
  F.get(0, nph, nuu, nww, &nd);
  nb = x[0];
  for (int i = 0; i < P.L; ++i) {
    const int n = n0 + i;
    if (n >= P.N) break;  // (padding: only the last chunk's lanes; wave-divergent tail, a few steps)
    if (n == P.N - 1) {   // the last sample: its transition is 0
      if (WITH_M) {
#pragma unroll
        for (int k = 0; k < J * J; ++k) M[k] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) a[j] = 0.0;
      break;
    }
    double ph[J], uu[J], ww[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { ph[j] = nph[j]; uu[j] = nuu[j]; ww[j] = nww[j]; }
    const double bn = nb;
    if (i + 1 < P.L && n + 1 < P.N) {
      F.get(i + 1, nph, nuu, nww, &nd);
      nb = x[(long)(i + 1) * P.nchunk];
    }
    // offset: x_n = b_n - u . a ; a <- phi (a + W x_n)
    double ua = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) ua = fma(uu[j], a[j], ua);
    const double xn = bn - ua;
#pragma unroll
    for (int j = 0; j < J; ++j) a[j] = ph[j] * fma(ww[j], xn, a[j]);
    if (WITH_M) {  // M <- Phi (M - W (u^T M))
      double pw[J];
#pragma unroll
      for (int j = 0; j < J; ++j) pw[j] = ph[j] * ww[j];
#pragma unroll
      for (int k = 0; k < J; ++k) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) r = fma(uu[j], M[j * J + k], r);
#pragma unroll
        for (int j = 0; j < J; ++j) M[j * J + k] = fma(-pw[j], r, ph[j] * M[j * J + k]);
      }
    }
  }
  const long slot = (long)b * P.nchunk + c;
  if (WITH_M) {
    double* o = S.M + slot * (J * J);
#pragma unroll
    for (int k = 0; k < J * J; ++k) o[k] = M[k];
  }
  double* oa = S.off + (((long)b * S.nrhs + S.r) * P.nchunk + c) * J;
#pragma unroll
  for (int j = 0; j < J; ++j) oa[j] = a[j];
}

// 2. / 4b. chunk start states of one sweep: one lane per (problem, right-hand side) walks the chunks
//    forward:  start[c] = g ; g <- M_c g + a_c        backward: start[c] = k ; k <- M_c^T k + off_c  (c descending)
template <int J, bool BACKWARD>
__global__ void __launch_bounds__(64) bsolve_prefix_kernel(const BatchParams P, const BSolveParams S) {
  const long idx = (long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long)P.B * S.nrhs) return;
  const int b = (int)(idx / S.nrhs);
  double g[J];
#pragma unroll
  for (int j = 0; j < J; ++j) g[j] = 0.0;
  for (int q = 0; q < P.nchunk; ++q) {
    const int c = BACKWARD ? P.nchunk - 1 - q : q;
    const double* M = S.M + ((long)b * P.nchunk + c) * (J * J);
    const double* a = S.off + (idx * P.nchunk + c) * J;
    double* st = S.starts + (idx * P.nchunk + c) * J;
    double nx[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { st[j] = g[j]; nx[j] = a[j]; }
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
      for (int k = 0; k < J; ++k) nx[j] = fma(BACKWARD ? M[k * J + j] : M[j * J + k], g[k], nx[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = nx[j];
  }
}

// 3. forward recurrence per chunk from its start state; x / D in place
template <int JR, int JC, bool LEAN, bool FAST>
__global__ void __launch_bounds__(64) bsolve_forward_kernel(const BatchParams P, const BSolveParams S) {
  constexpr int J = JR + 2 * JC;
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const auto F = make_slots<JR, JC, LEAN, FAST>(P, b, c);
  const long cells = (long)P.L * P.nchunk;
  double* x = S.xT + ((long)b * S.nrhs + S.r) * cells + c;
  const double* st = S.starts + (((long)b * S.nrhs + S.r) * P.nchunk + c) * J;
  double g[J];
#pragma unroll
  for (int j = 0; j < J; ++j) g[j] = st[j];
  const int n0 = c * P.L;
  double nph[J], nuu[J], nww[J], nd, nb;
  F.get(0, nph, nuu, nww, &nd);
  nb = x[0];
  for (int i = 0; i < P.L; ++i) {
    const int n = n0 + i;
    if (n >= P.N) break;
    double ph[J], uu[J], ww[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { ph[j] = nph[j]; uu[j] = nuu[j]; ww[j] = nww[j]; }
    const double d = nd, bn = nb;
    if (i + 1 < P.L && n + 1 < P.N) {  // (the next step's slot, one step ahead)
      F.get(i + 1, nph, nuu, nww, &nd);
      nb = x[(long)(i + 1) * P.nchunk];
    }
    double ug = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) ug = fma(uu[j], g[j], ug);
    const double xn = bn - ug;
    x[(long)i * P.nchunk] = xn / d;
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = ph[j] * fma(ww[j], xn, g[j]);
  }
}

// 4a. / 5. backward recurrence per chunk: OFFSETS: from k = 0, the chunk's backward offset (x untouched); else from the
// chunk's start state, x in place
template <int JR, int JC, bool LEAN, bool FAST, bool OFFSETS>
__global__ void __launch_bounds__(64) bsolve_backward_kernel(const BatchParams P, const BSolveParams S) {
  constexpr int J = JR + 2 * JC;
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const auto F = make_slots<JR, JC, LEAN, FAST>(P, b, c);
  const long cells = (long)P.L * P.nchunk;
  double* x = S.xT + ((long)b * S.nrhs + S.r) * cells + c;
  double k[J];
  if (OFFSETS) {
#pragma unroll
    for (int j = 0; j < J; ++j) k[j] = 0.0;
  } else {
    const double* st = S.starts + (((long)b * S.nrhs + S.r) * P.nchunk + c) * J;
#pragma unroll
    for (int j = 0; j < J; ++j) k[j] = st[j];
  }
  const int n0 = c * P.L;
  const int last = (P.N - n0 < P.L) ? P.N - n0 : P.L;  // samples of this chunk inside the series
  double nph[J], nuu[J], nww[J], nd, nb;
  F.get(last - 1, nph, nuu, nww, &nd);
  nb = x[(long)(last - 1) * P.nchunk];
  for (int i = last - 1; i >= 0; --i) {
    const int n = n0 + i;
    double ph[J], uu[J], ww[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { ph[j] = nph[j]; uu[j] = nuu[j]; ww[j] = nww[j]; }
    double xn = nb;
    if (i > 0) {  // (the previous sample's slot, one step ahead)
      F.get(i - 1, nph, nuu, nww, &nd);
      nb = x[(long)(i - 1) * P.nchunk];
    }
    if (n == P.N - 1) {  // the reference starts its backward sweep with f = 0
#pragma unroll
      for (int j = 0; j < J; ++j) k[j] = uu[j] * xn;
      continue;
    }
    double h[J], wh = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) { h[j] = ph[j] * k[j]; wh = fma(ww[j], h[j], wh); }
    // (OFFSETS: the offset of the affine map -- the same recurrence with the state's own contribution left out of x:
    //  x_n = xf_n - W . h holds for the offset part h of the state as well, since the map is affine in k)
    xn -= wh;
    if (!OFFSETS) x[(long)i * P.nchunk] = xn;
#pragma unroll
    for (int j = 0; j < J; ++j) k[j] = fma(uu[j], xn, h[j]);
  }
  if (OFFSETS) {
    double* o = S.off + (((long)b * S.nrhs + S.r) * P.nchunk + c) * J;
#pragma unroll
    for (int j = 0; j < J; ++j) o[j] = k[j];
  }
}

}  // namespace clr
