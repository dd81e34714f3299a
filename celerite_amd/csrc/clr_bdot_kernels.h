// celerite_amd/csrc/clr_bdot_kernels.h -- K z for every problem of a plan: the batched form of CholeskySolver::dot
// (cholesky.h:441-596; GP.dot, celerite.py:453-489), parallel in n (round 6).  No factor is needed: K is given by the
// plan's resident times and the coefficients in force; its diagonal is sum a_real + sum a_comp + jitter (:483-485 -- the
// observational variance is NOT part of it).
//
// The reference runs two diagonal recurrences per right-hand side (its phi(:, n) = the decay n -> n+1,
// u(:, n) = U~(t_{n+1}), v(:, n) = V~(t_n)):
//     upper triangle, n = N-2 .. 0:   f <- phi_n (f + U~(t_{n+1}) z_{n+1}) ;  y_n  = diag z_n + V~(t_n) . f      :536-547
//     lower triangle, n = 1 .. N-1:   f <- phi_{n-1} (f + V~(t_{n-1}) z_{n-1}) ;  y_n += U~(t_n) . f             :549-559
// Per sample (one evaluation of U~, V~ at t_n and of the decay t_n -> t_{n+1}):
//     PASS 0 (n descending):  h = phi_n k ;  y_n = diag z_n + V~_n . h ;  k <- h + U~_n z_n      k = what sample n - 1 reads
//     PASS 1 (n ascending):   y_n += U~_n . g ;  g <- phi_n (g + V~_n z_n)                        g = what sample n + 1 reads
// Both transitions are diagonal, so a chunk maps its incoming state to its outgoing one as p (.) s + a with p = the
// product of the chunk's decays: per pass the three phases of clr_bdotl_kernels.h (offsets and decay products, a walk over
// the chunks, the recurrence from the start states), lane = (problem, chunk), z and y in the chunk-interleaved layout
// ([problem][rhs][i][chunk]).  The last sample has no successor: its decay is never formed.
#pragma once

namespace clr {

struct BDotParams {
  int nrhs;
  const double* zT;     // [B][nrhs][L][nchunk]
  double* yT;           // [B][nrhs][L][nchunk]
  double* decay;        // [B][nchunk][J]      the chunks' decay products (the same for both passes)
  double* off;          // [B][nrhs][nchunk][J] chunk offsets of the running pass
  double* starts;       // [B][nrhs][nchunk][J] chunk start states of the running pass
};

template <int JR, int JC, bool FAST, int PASS, bool REPLAY>
__global__ void __launch_bounds__(64) bdot_kernel(const BatchParams P, const BDotParams S) {
  constexpr int J = JR + 2 * JC;
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.z;
  if (c >= P.nchunk) return;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  const DirectSeries ts{P.t + b * P.t_stride + c * P.lane_cs, nullptr, nullptr, P.lane_is, P.lane_cs, P.L, (long)P.N - (long)c * P.L};
  const long cells = (long)P.L * P.nchunk;
  const double* z = S.zT + ((long)b * S.nrhs + r) * cells + c;
  double* y = S.yT + ((long)b * S.nrhs + r) * cells + c;
  const long slot = ((long)b * S.nrhs + r) * P.nchunk + c;
  const double dg = (p.sum_ar + p.sum_ac) + p.jitter;  // :483
  double st[J], pd[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { st[j] = REPLAY ? S.starts[slot * J + j] : 0.0; pd[j] = 1.0; }
  const int n0 = c * P.L;
  const int last = (P.N - n0 < P.L) ? P.N - n0 : P.L;  // samples of this chunk inside the series (>= 1)
  for (int q = 0; q < last; ++q) {
    const int i = PASS == 0 ? last - 1 - q : q;
    const int n = n0 + i;
    const double tn = ts.t(i), zn = z[(long)i * P.nchunk];
    double uu[J], vv[J], ph[J];
    features_uv<JR, JC, FAST>(p, tn, uu, vv);
    const bool tail = n == P.N - 1;
    if (!tail) {
      double phid[nz(JR + JC)];
      features_phi_distinct<JR, JC>(p, ts.t(i + 1) - tn, phid);
#pragma unroll
      for (int j = 0; j < J; ++j) ph[j] = phid[phi_index<JR>(j)];
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) ph[j] = 0.0;
    }
    if (PASS == 0) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double h = ph[j] * st[j];  // (the last sample: k = 0 comes in and phi is defined as 0)
        acc = fma(vv[j], h, acc);
        st[j] = fma(uu[j], zn, h);
        if (!REPLAY) pd[j] *= ph[j];
      }
      if (REPLAY) y[(long)i * P.nchunk] = fma(dg, zn, acc);
    } else {
      if (REPLAY) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) acc = fma(uu[j], st[j], acc);
        y[(long)i * P.nchunk] += acc;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        st[j] = ph[j] * fma(vv[j], zn, st[j]);  // (the last sample: its outgoing state is never read)
        if (!REPLAY) pd[j] *= ph[j];
      }
    }
  }
  if (!REPLAY) {
#pragma unroll
    for (int j = 0; j < J; ++j) S.off[slot * J + j] = st[j];
    if (r == 0 && PASS == 0) {
      double* o = S.decay + ((long)b * P.nchunk + c) * J;
#pragma unroll
      for (int j = 0; j < J; ++j) o[j] = pd[j];
    }
  }
}

// one lane per (problem, right-hand side) walks the chunks (PASS 0: downwards): start[c] = s ; s <- p_c s + a_c
template <int J, int PASS>
__global__ void __launch_bounds__(64) bdot_prefix_kernel(const BatchParams P, const BDotParams S) {
  const long idx = (long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long)P.B * S.nrhs) return;
  const int b = (int)(idx / S.nrhs);
  double s[J];
#pragma unroll
  for (int j = 0; j < J; ++j) s[j] = 0.0;
  for (int q = 0; q < P.nchunk; ++q) {
    const int c = PASS == 0 ? P.nchunk - 1 - q : q;
    const double* pd = S.decay + ((long)b * P.nchunk + c) * J;
    const double* a = S.off + (idx * P.nchunk + c) * J;
    double* o = S.starts + (idx * P.nchunk + c) * J;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      o[j] = s[j];
      s[j] = fma(pd[j], s[j], a[j]);
    }
  }
}

}  // namespace clr
