// celerite_amd/csrc/clr_bdotl_kernels.h -- L z for every problem of a plan from its materialised factor: the batched form
// of CholeskySolver::dot_L (cholesky.h:409-431; GP.sample, celerite.py:422-451), parallel in n (round 6).
//
// The reference runs, per right-hand side (phi_, u_, W_, D_ in its storage),
//     f <- phi_{n-1} (f + W_{n-1} sqrt(D_{n-1}) z_{n-1}) ;  y_n = sqrt(D_n) z_n + u_{n-1} . f          n = 1 .. N-1
// which, in the factor's own SLOTS (clr_bsolve_kernels.h: slot n holds phi[n] = the decay n -> n+1, u[n] = U~(t_n),
// W[n], D[n]), touches one slot per sample:
//     y_n = sqrt(D[n]) z_n + u[n] . g ;  g <- phi[n] (g + W[n] sqrt(D[n]) z_n)                          g = the f the NEXT sample reads
// The state's transition is DIAGONAL (phi[n]): a chunk maps its start state to its end state as g -> p g + a with
// p = the product of the chunk's decays and a = the state it reaches from zero.  Three phases, lane = (problem, chunk):
//   1. bdotl_kernel<.., false>   per chunk p (shared by all right-hand sides) and a (no u, no y: 2 J + 2 doubles per sample);
//   2. bdotl_prefix_kernel       per (problem, right-hand side) the state every chunk starts from: a walk over the chunks;
//   3. bdotl_kernel<.., true>    the recurrence per chunk from its start state: y in place of z.
// The right-hand sides live in the chunk-interleaved layout of the series ([problem][rhs][i][chunk]).  LEAN: phi, u are
// regenerated per step from the times and the coefficients (FactorSlots).  The last sample has no successor: its
// transition is defined as 0 (never read); padded samples n >= N are skipped.
#pragma once

namespace clr {

struct BDotLParams {
  int nrhs;
  int lean;             // the factor holds W, D only
  double* xT;           // [B][nrhs][L][nchunk] z in, L z out
  double* decay;        // [B][nchunk][J] the chunks' decay products
  double* off;          // [B][nrhs][nchunk][J] chunk offsets
  double* starts;       // [B][nrhs][nchunk][J] chunk start states
};

template <int JR, int JC, bool LEAN, bool FAST, bool REPLAY>
__global__ void __launch_bounds__(64) bdotl_kernel(const BatchParams P, const BDotLParams S) {
  constexpr int J = JR + 2 * JC;
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.z;
  if (c >= P.nchunk) return;
  const auto F = make_slots<JR, JC, LEAN, FAST>(P, b, c);
  const long cells = (long)P.L * P.nchunk;
  double* x = S.xT + ((long)b * S.nrhs + r) * cells + c;
  const long slot = ((long)b * S.nrhs + r) * P.nchunk + c;
  double g[J], pd[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { g[j] = REPLAY ? S.starts[slot * J + j] : 0.0; pd[j] = 1.0; }
  const int n0 = c * P.L;
  // (the slot of step i + 1 and its z one step ahead of the arithmetic, as in the batched solve)
  double nph[J], nuu[J], nww[J], nd, nz;
  F.get(0, nph, nuu, nww, &nd);
  nz = x[0];
  for (int i = 0; i < P.L; ++i) {
    const int n = n0 + i;
    if (n >= P.N) break;  // (padding: only the last chunk's lanes)
    double ph[J], uu[J], ww[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { ph[j] = nph[j]; uu[j] = nuu[j]; ww[j] = nww[j]; }
    const double tz = sqrt(nd) * nz;  // :421, :426
    if (i + 1 < P.L && n + 1 < P.N) {
      F.get(i + 1, nph, nuu, nww, &nd);
      nz = x[(long)(i + 1) * P.nchunk];
    }
    if (REPLAY) {
      double ug = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) ug = fma(uu[j], g[j], ug);
      x[(long)i * P.nchunk] = tz + ug;  // :427
    }
    if (n == P.N - 1) {  // the last sample: no successor
#pragma unroll
      for (int j = 0; j < J; ++j) { g[j] = 0.0; pd[j] = 0.0; }
      break;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      g[j] = ph[j] * fma(ww[j], tz, g[j]);  // :425
      if (!REPLAY) pd[j] *= ph[j];
    }
  }
  if (!REPLAY) {
#pragma unroll
    for (int j = 0; j < J; ++j) S.off[slot * J + j] = g[j];
    if (r == 0) {
      double* o = S.decay + ((long)b * P.nchunk + c) * J;
#pragma unroll
      for (int j = 0; j < J; ++j) o[j] = pd[j];
    }
  }
}

// 2. one lane per (problem, right-hand side) walks the chunks: start[c] = g ; g <- p_c g + a_c
template <int J>
__global__ void __launch_bounds__(64) bdotl_prefix_kernel(const BatchParams P, const BDotLParams S) {
  const long idx = (long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long)P.B * S.nrhs) return;
  const int b = (int)(idx / S.nrhs);
  double g[J];
#pragma unroll
  for (int j = 0; j < J; ++j) g[j] = 0.0;
  for (int c = 0; c < P.nchunk; ++c) {
    const double* pd = S.decay + ((long)b * P.nchunk + c) * J;
    const double* a = S.off + (idx * P.nchunk + c) * J;
    double* st = S.starts + (idx * P.nchunk + c) * J;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      st[j] = g[j];
      g[j] = fma(pd[j], g[j], a[j]);
    }
  }
}

}  // namespace clr
