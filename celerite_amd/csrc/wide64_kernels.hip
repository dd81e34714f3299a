// celerite_amd/csrc/wide64_kernels.hip -- the chunk algebra of the wide scan at the padded width 64 (widths 33..64;
// round 5): the prefix over a problem's chunks AND the corrections of every chunk in ONE walk.
//
// The loop being parallelised is cholesky.h:126-179 fused with :348-357; summarize (wide_scan_body, MODE 1) has turned
// every chunk into its element (A, b, C, eta, Jm) and its zero-start sums.  At widths <= 32 two
// kernels follow: the prefix (start state of every chunk: wide_prefix32_kernel / prefix_coop_kernel) and
// wide_correct_kernel (true log det / quadratic contributions + the positive-definiteness certificate from (start
// state, element)).  Both eliminate the SAME tableau
//     T = [ I + P Jm | P | f + P eta ]  --Gauss-Jordan, partial pivoting-->  [ I | G | g ],  det(I + P Jm)
// with P, f the chunk's start state: the advance needs G = (I + P Jm)^-1 P and g = (I + P Jm)^-1 (f + P eta),
//     P' = C + A sym(G) A^T ,  f' = A g + b ,
// the corrections need det, G and w = Jm f - eta (determinant lemma + Woodbury):
//     log det += log det(I + P Jm) ,  quad += 2 eta.f - f.Jm f + w.sym(G) w ,
// and the certificate two Cholesky factorisations: F F^T = -Jm + delta I, smallest pivot mu of I - F^T P F > 1e-5.
// At width 64 one such tableau is 66 KB: a workgroup of 256 threads per PROBLEM walks its chunks with three 64 x 64
// buffers in LDS (132 KB; the element's A and C are read from L2) and does both jobs per chunk -- one elimination
// instead of two, two launches fewer.  Plain FMA loops: with <= 16 chunks per problem the walk is a few hundred
// microseconds beside a summarize pass of tens of milliseconds.
// Same thresholds and the same outputs as wide_correct_kernel / chunk_update (clr_core.h): part, cond (mu, error
// estimate), egerr (measured accuracy of G on two probe vectors), flags / need_exact.
#include "../../include/celerite_hip.h"
#include "clr_batch_kernels.h"
#include "clr_wide.h"

namespace clr {

namespace {

// PHASE 0: the walk does everything (few chunks per problem: the corrections ride on the elimination the advance needs anyway).
// PHASE 1 / 2 (many chunks per problem -- one long series through CholeskySolver, small batches): the walk only ADVANCES
// (elimination + the two products: the dependent chain), and a second launch, one workgroup per (problem, chunk), repeats
// the elimination from the start state the walk stored and does the corrections, the probes and the certificate in
// parallel over the chunks -- the arrangement of the widths <= 32 (wide_prefix32_kernel + wide_correct_kernel).
template <int J, int PHASE>
__global__ void __launch_bounds__(256) wide_walk_kernel(const BatchParams P) {
  constexpr bool ADVANCE = PHASE != 2, CORRECT = PHASE != 1;
  constexpr int SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  constexpr int LD = J + 1, NC = 2 * J + 1, LT = NC + 1, NT = 256;
  extern __shared__ double lds[];
  double* Pm = lds;                 // [J][LD]  the start state P; later E = I - F^T P F (certificate)
  double* Jf = Pm + J * LD;         // [J][LD]  Jm; later -Jm + delta I -> F (lower triangle); later sym(G)
  double* T = Jf + J * LD;          // [J][LT]  the tableau; left half later P F, then X = A sym(G)
  double* fv = T + J * LT;          // [J] f
  double* ev = fv + J;              // [J] eta
  double* wv = ev + J;              // [J] w = Jm f - eta
  double* pa = wv + J;              // [J] scratch
  double* pb = pa + J;              // [J] scratch
  const int tid = threadIdx.x, lane = tid & 63, b = PHASE == 2 ? blockIdx.y : blockIdx.x;
  const long slot0 = (long)b * P.nchunk;
  if (PHASE != 2 && tid == 0) {
    int raised = 0;
    for (int c = 0; c < P.nchunk; ++c) raised |= P.flags[slot0 + c] ? 2 : 0;  // a zero-start pivot <= 0 (summarize)
    P.need_exact[b] = raised;
    if (P.egerr) P.egerr[slot0] = 0.0;  // the first chunk starts from the zero state: nothing to correct
  }
  if (PHASE == 2) {  // this workgroup's chunk: its start state as the walk stored it
    const double* st = P.starts + (slot0 + blockIdx.x + 1) * START;
    for (int idx = tid; idx < J * J; idx += NT) Pm[(idx / J) * LD + idx % J] = st[sym(idx / J, idx % J)];
    if (tid < J) fv[tid] = st[SZ + tid];
  } else {  // start state of chunk 1 = (C, b) of chunk 0 (its zero-start trajectory)
    const double* E = P.elems + slot0 * ELEM;
    const double* EC = E + J * J + J;
    for (int idx = tid; idx < J * J; idx += NT) Pm[(idx / J) * LD + idx % J] = EC[sym(idx / J, idx % J)];
    if (tid < J) fv[tid] = E[J * J + tid];
  }
  __syncthreads();
  const int c_lo = PHASE == 2 ? (int)blockIdx.x + 1 : 1, c_hi = PHASE == 2 ? c_lo + 1 : P.nchunk;
  for (int c = c_lo; c < c_hi; ++c) {
    if (PHASE == 1 && c + 1 == P.nchunk) {  // (advance only: the last chunk's start state, nothing to advance through)
      double* o = P.starts + (slot0 + c) * START;
      for (int idx = tid; idx < J * J; idx += NT) {
        const int i = idx / J, j = idx % J;
        if (i <= j) o[tri(i, j)] = Pm[i * LD + j];
      }
      if (tid < J) o[SZ + tid] = fv[tid];
      break;
    }
    const long slot = slot0 + c;
    const double* E = P.elems + slot * ELEM;
    const double* Eb = E + J * J;
    const double* EC = Eb + J;
    const double* Eeta = EC + SZ;
    const double* EJm = Eeta + J;
    {  // the chunk's start state, packed upper triangle | f
      double* o = P.starts + slot * START;
      for (int idx = tid; idx < J * J; idx += NT) {
        const int i = idx / J, j = idx % J;
        if (PHASE != 2 && i <= j) o[tri(i, j)] = Pm[i * LD + j];
        Jf[i * LD + j] = EJm[sym(i, j)];
      }
      if (tid < J) { if (PHASE != 2) o[SZ + tid] = fv[tid]; ev[tid] = Eeta[tid]; }
    }
    __syncthreads();
    // T = [ I + P Jm | P | f + P eta ]
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      double acc = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < J; ++k) acc += Pm[i * LD + k] * Jf[k * LD + j];
      T[i * LT + j] = acc;
      T[i * LT + J + j] = Pm[i * LD + j];
    }
    if (tid < J) {
      double h = fv[tid];
      for (int j = 0; j < J; ++j) h += Pm[tid * LD + j] * ev[j];
      T[tid * LT + 2 * J] = h;
    }
    __syncthreads();
    // Gauss-Jordan, partial pivoting.  Thread = (column cc of [M^T | P], half of the rows); the first J threads also carry
    // the extra column (f + P eta).  Per pivot: search (every wave), read the two rows, barrier, swap + scale, barrier,
    // eliminate (the multipliers T[i][col] stay in place: columns <= col are never updated), barrier.
    double det = 1.0;
    const int cc = tid & 127, rg = tid >> 7;
    for (int col = 0; col < J; ++col) {
      // pivot row: the largest |T[i][col]|, i >= col (first one on ties, as the single-lane scan); every wave searches
      double best = (lane < J && lane >= col) ? fabs(T[lane * LT + col]) : -1.0;
      int piv = lane;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const double ob = __shfl_xor(best, m, 64);
        const int op = __shfl_xor(piv, m, 64);
        const bool take = ob > best || (ob == best && op < piv);
        best = take ? ob : best;
        piv = take ? op : piv;
      }
      const double p = T[piv * LT + col];
      det *= (piv != col) ? -p : p;
      const double inv = 1.0 / p;
      const bool mine = cc < 2 * J;
      const double top = mine ? T[piv * LT + cc] : 0.0, old = mine ? T[col * LT + cc] : 0.0;
      const double topx = T[piv * LT + 2 * J], oldx = T[col * LT + 2 * J];
      const double t = top * inv, tx = topx * inv;
      __syncthreads();  // (everyone has read rows piv and col)
      if (rg == 0 && mine) {
        T[piv * LT + cc] = old;                         // the row swap (a no-op when piv == col) ...
        T[col * LT + cc] = (cc > col) ? t : top;        // ... and the scaled pivot row
      }
      if (tid == 0) { T[piv * LT + 2 * J] = oldx; }
      __syncthreads();
      if (tid == 0) T[col * LT + 2 * J] = tx;           // (after the barrier: piv == col must end with the scaled value)
      if (mine && cc > col) {
        for (int i = rg; i < J; i += 2)
          if (i != col) T[i * LT + cc] = fma(-T[i * LT + col], t, T[i * LT + cc]);
      }
      if (tid < J && tid != col) T[tid * LT + 2 * J] = fma(-T[tid * LT + col], tx, T[tid * LT + 2 * J]);
      __syncthreads();
    }
    // T[i][J + j] = G[i][j], T[i][2J] = g[i]

    if (CORRECT) {
    // measured accuracy of G (chunk_update's eg_out, the same two probe vectors)
    double eg = 0.0;
    if (P.egerr) {
      for (int probe = 0; probe < 2; ++probe) {
        double gz = 0.0, r = 0.0;
        if (tid < J) {
          for (int k = 0; k < J; ++k) gz += (probe && (k & 1)) ? -T[tid * LT + J + k] : T[tid * LT + J + k];
          pa[tid] = gz;
        }
        __syncthreads();
        if (tid < J) {  // t1 = Jm (G z)
          double acc = 0.0;
          for (int k = 0; k < J; ++k) acc += Jf[tid * LD + k] * pa[k];
          pb[tid] = acc;
        }
        __syncthreads();
        if (tid < J) {  // r = P (z - t1) - G z
          double acc = -gz;
          for (int k = 0; k < J; ++k) acc += Pm[tid * LD + k] * (((probe && (k & 1)) ? -1.0 : 1.0) - pb[k]);
          r = acc;
          pa[tid] = r;
        }
        __syncthreads();
        if (tid < J) {  // t1 = Jm r
          double acc = 0.0;
          for (int k = 0; k < J; ++k) acc += Jf[tid * LD + k] * pa[k];
          pb[tid] = acc;
        }
        __syncthreads();
        double dg = 0.0;
        if (tid < J) {  // dg = r - G t1
          double acc = r;
          for (int k = 0; k < J; ++k) acc -= T[tid * LT + J + k] * pb[k];
          dg = (acc != acc) ? INFINITY : fabs(acc);
        }
        double gmax = (tid < J) ? fabs(gz) : 0.0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {  // (rows live in the first wave: tid = lane there; the other waves hold zeros)
          dg = fmax(dg, __shfl_xor(dg, m, 64));
          gmax = fmax(gmax, __shfl_xor(gmax, m, 64));
        }
        const double e = (gmax > 0.0) ? dg / gmax : (dg == 0.0 ? 0.0 : INFINITY);
        eg = (e > eg || e != e) ? e : eg;
        __syncthreads();
      }
    }

    // w = Jm f - eta ;  eta.f ;  f.Jm f ;  w.sym(G) w
    double ef = 0.0, fJf = 0.0, wGw = 0.0;
    if (tid < J) {
      double acc = 0.0;
      for (int k = 0; k < J; ++k) acc += Jf[tid * LD + k] * fv[k];
      wv[tid] = acc - ev[tid];
      ef = ev[tid] * fv[tid];
      fJf = fv[tid] * acc;
    }
    __syncthreads();
    if (tid < J) {
      double acc = 0.0;
      for (int k = 0; k < J; ++k) acc += 0.5 * (T[tid * LT + J + k] + T[k * LT + J + tid]) * wv[k];
      wGw = wv[tid] * acc;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {  // (first wave: the J rows; J <= 64)
      ef += __shfl_xor(ef, m, 64);
      fJf += __shfl_xor(fJf, m, 64);
      wGw += __shfl_xor(wGw, m, 64);
    }

    // certificate: F F^T = N + delta I (N = -Jm) in place of Jm, P F in the tableau's left half, E = I - F^T P F in
    // place of P (the start state has been written out and the advance below does not read it)
    double nmax = (lane < J) ? -Jf[lane * LD + lane] : 0.0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) nmax = fmax(nmax, __shfl_xor(nmax, m, 64));
    const double delta = 4e-13 * nmax;
    __syncthreads();  // (every reader of Jm above is done)
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      Jf[i * LD + j] = -Jf[i * LD + j] + ((i == j) ? delta : 0.0);
    }
    __syncthreads();
    // right-looking Cholesky, thread = (row i, every fourth column): the trailing update with the UNSCALED column k
    // (F_ij -= F_ik F_jk / F_kk), one barrier, then column k is scaled in place -- nothing reads it again before the end
    {
      const int ci = tid & 63, cj = tid >> 6;
      for (int k = 0; k < J; ++k) {
        const double dk = Jf[k * LD + k];
        const double ivk = 1.0 / dk;
        if (ci < J && ci > k) {
          const double fik = Jf[ci * LD + k] * ivk;
          for (int j = k + 1 + cj; j <= ci; j += 4) Jf[ci * LD + j] = fma(-fik, Jf[j * LD + k], Jf[ci * LD + j]);
        }
        __syncthreads();
        if (cj == 0 && ci < J) Jf[ci * LD + k] = (ci >= k) ? Jf[ci * LD + k] * (1.0 / sqrt(dk)) : 0.0;
      }
      __syncthreads();
    }
    for (int idx = tid; idx < J * J; idx += NT) {  // P F -> T[:, 0:J]
      const int i = idx / J, k = idx % J;
      double acc = 0.0;
      for (int m = k; m < J; ++m) acc += Pm[i * LD + m] * Jf[m * LD + k];
      T[i * LT + k] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {  // E = I - F^T (P F), lower triangle -> Pm
      const int j = idx / J, k = idx % J;
      if (k <= j) {
        double acc = (j == k) ? 1.0 : 0.0;
        for (int i = k; i < J; ++i) acc -= Jf[i * LD + k] * T[i * LT + j];
        Pm[j * LD + k] = acc;
      }
    }
    __syncthreads();
    double mu = 1.0;
    bool broke = false;
    {
      const int ci = tid & 63, cj = tid >> 6;
      for (int k = 0; k < J; ++k) {  // (only the pivots are wanted: the scaled columns are never formed)
        const double dk = Pm[k * LD + k];
        if (!(dk > 0.0)) broke = true;
        mu = (dk < mu) ? dk : mu;
        const double ivk = 1.0 / dk;
        if (ci < J && ci > k) {
          const double fik = Pm[ci * LD + k] * ivk;
          for (int j = k + 1 + cj; j <= ci; j += 4) Pm[ci * LD + j] = fma(-fik, Pm[j * LD + k], Pm[ci * LD + j]);
        }
        __syncthreads();
      }
    }
    if (broke) mu = -1.0;

    if (tid == 0) {
      int bad = 0;
      if (!(mu > 1e-5)) bad = 1;
      if (!(det > 0.0)) bad = 1;
      const double ld0 = P.part[slot * 2 + 0], q0 = P.part[slot * 2 + 1];
      const double q = 2.0 * ef - fJf + wGw;
      const double ld = log(det);
      const double err = J * 2.2e-16 / mu;  // rounding-error estimate of the corrections (decide_kernel sums them)
      if ((!P.logdet_only && !isfinite(q)) || !isfinite(ld)) bad = 1;
      P.part[slot * 2 + 0] = ld0 + ld;
      P.part[slot * 2 + 1] = q0 + q;
      if (P.cond) { P.cond[slot * 3 + 1] = mu; P.cond[slot * 3 + 2] = P.logdet_only ? 0.0 : err * fabs(wGw); }
      if (P.egerr) P.egerr[slot] = eg;
      if (bad) {
        P.flags[slot] |= 2;
        atomicOr(P.need_exact + b, 2);
      }
    }
    }  // CORRECT
    if (!ADVANCE) break;
    if (c + 1 == P.nchunk) break;  // (the last chunk is corrected, not advanced through)

    // advance: sym(G) -> Jf ; X = A sym(G) -> T[:, 0:J] ; P' = C + X A^T -> Pm ; f' = A g + b
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      Jf[i * LD + j] = 0.5 * (T[i * LT + J + j] + T[j * LT + J + i]);
    }
    if (tid < J) pa[tid] = T[tid * LT + 2 * J];  // g
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      double acc = 0.0;
      for (int k = 0; k < J; ++k) acc += E[i * J + k] * Jf[k * LD + j];
      T[i * LT + j] = acc;
    }
    if (tid < J) {
      double acc = Eb[tid];
      for (int k = 0; k < J; ++k) acc += E[tid * J + k] * pa[k];
      fv[tid] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (i <= j) {  // (the upper triangle is computed, the lower one mirrored: the state stays exactly symmetric)
        double acc = EC[tri(i, j)];
        for (int k = 0; k < J; ++k) acc += T[i * LT + k] * E[j * J + k];
        Pm[i * LD + j] = acc;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (i > j) Pm[i * LD + j] = Pm[j * LD + i];
    }
    __syncthreads();
  }
}

template <int J>
size_t walk_lds_bytes() {
  return ((size_t)2 * J * (J + 1) + (size_t)J * (2 * J + 2) + 5 * (size_t)J) * sizeof(double);
}

}  // namespace

// prefix + corrections in one walk per problem (the padded width 64; 32 for cross-checks against the two-kernel path)
int launch_wide_walk(const BatchParams& P, int width_padded, hipStream_t s) {
  if (P.nchunk < 2) return 0;
  // (more than the default 64 KB of LDS per workgroup at width 64: asked for on every device the library runs on)
  // few chunks: one fused walk; many (one long series, small batches): the walk advances, the corrections run in parallel
  // (measured, profiles/r05i_wide64_split_walk.txt: a fused walk step 0.45 ms, an advance-only one 0.27 ms, a round of 256
  //  correction workgroups -- 132 KB of LDS each: one per CU -- 0.45 ms)
  const long nc1 = P.nchunk - 1, rounds = ((long)P.B * nc1 + 255) / 256;
  const bool split = 0.27 * nc1 + 0.45 * rounds < 0.45 * nc1;
  const dim3 cgrid(P.nchunk - 1, P.B);
  if (width_padded <= 32) {
    if (!split) hipLaunchKernelGGL((wide_walk_kernel<32, 0>), dim3(P.B), dim3(256), walk_lds_bytes<32>(), s, P);
    else {
      hipLaunchKernelGGL((wide_walk_kernel<32, 1>), dim3(P.B), dim3(256), walk_lds_bytes<32>(), s, P);
      hipLaunchKernelGGL((wide_walk_kernel<32, 2>), cgrid, dim3(256), walk_lds_bytes<32>(), s, P);
    }
  } else {
    const int bytes = (int)walk_lds_bytes<64>();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_walk_kernel<64, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_walk_kernel<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_walk_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return 1;
    if (!split) hipLaunchKernelGGL((wide_walk_kernel<64, 0>), dim3(P.B), dim3(256), bytes, s, P);
    else {
      hipLaunchKernelGGL((wide_walk_kernel<64, 1>), dim3(P.B), dim3(256), bytes, s, P);
      hipLaunchKernelGGL((wide_walk_kernel<64, 2>), cgrid, dim3(256), bytes, s, P);
    }
  }
  return 0;
}

}  // namespace clr
