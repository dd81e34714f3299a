// celerite_amd/csrc/bigsweep_kernels.hip -- dot_solve (cholesky.h:343-357) and solve (:236-260) over a stored factor at
// widths 65 .. 1024 as CHUNKED AFFINE SCANS along n (round 6; wsweep_kernels.hip keeps a chunk's whole (J + 1)^2 map in one
// wave's registers and stops at width 64; above it the sweeps walked the series with one wave / one workgroup per
// right-hand side: 0.23 .. 1.7 us per sample, 0.1 - 0.35x one CPU core).
//
// Forward:  f_n = phi_{n-1} o (f_{n-1} + W_{n-1} x_{n-1}),  x_n = b_n - u_{n-1} . f_n   is, in f alone,
//           f_n = H_n f_{n-1} + phi_{n-1} o W_{n-1} b_{n-1},   H_n = diag(phi_{n-1}) (I - W_{n-1} u_{n-2}^T)      (n >= 2)
// Backward: f_n = phi_n o (f_{n+1} + u_n x_{n+1}),  x_n = z_n - W_n . f_n   is
//           f_n = G_n f_{n+1} + phi_n o u_n z_{n+1},           G_n = diag(phi_n) (I - u_n W_{n+1}^T)                (n <= N - 3)
// so a chunk of steps maps its start state affinely to its end state: f_end = A_c f_start + a_c(b).
//   maps     A_c = the product of the chunk's H (G), J x J, depends on the FACTOR only: built once per factor and direction
//            (bigsweep_maps_kernel), kept by the solver, shared by every right-hand side and every later call.  Its
//            columns evolve independently, so a workgroup takes a block of columns of one chunk: 512 threads x 32 entries
//            in registers, per step the column sums t = u^T A (DPP over the 16 .. 64 lanes that share a column) and
//            A <- diag(phi) (A - W t).  nchunk x J / (16 .. 128) workgroups: the whole chip.
//   a_c      the chunk's recurrence from a ZERO start state with the real right-hand side (bigsweep_chunk_kernel<false>),
//            one workgroup per (chunk, right-hand side), thread per row, one barrier per step;
//   walk     F_{c+1} = A_c F_c + a_c over the chunks (bigsweep_walk_kernel), one workgroup per right-hand side;
//   replay   the chunk again from its true start state F_c (bigsweep_chunk_kernel<true>): x, and dot_solve's sum.
// Storage (reference layout): phi, u: [j + J n], n = 0 .. N - 2 (decay n -> n + 1, u~(t_{n+1})); W: [j + J n], n = 0 .. N - 1.
#include <hip/hip_runtime.h>

#include <math.h>

#include <algorithm>

#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_wide.h"

namespace clr {

namespace {

constexpr int BIG_THREADS = 512;

// sum over the TR adjacent lanes that share a block of columns, delivered to all of them
template <int TR>
__device__ __forceinline__ double big_group_sum(double v) {
  v = dpp_add<DPP_QUAD_XOR1>(v);
  v = dpp_add<DPP_QUAD_XOR2>(v);
  v = dpp_add<DPP_HALF_MIRROR>(v);
  v = dpp_add<DPP_MIRROR>(v);
  if (TR >= 32) v = swap_add16(v);
  if (TR >= 64) v = swap_add32(v);
  return v;
}

// chunk c covers the steps s = c L + 1 .. min((c + 1) L, N - 1): forward step s is sample n = s (f_{n-1} -> f_n),
// backward step s is sample n = N - 1 - s (f_{n+1} -> f_n)
struct BigGeom {
  int N, J, nchunk, L;
};

// R rows x C columns per thread (R C = 32); TR = JP / R adjacent lanes cover all rows of a block of C columns
template <int R, int TR, bool BACKWARD>
__global__ void __launch_bounds__(BIG_THREADS) bigsweep_maps_kernel(BigGeom Gm, const double* __restrict__ phi, const double* __restrict__ u,
                                                                    const double* __restrict__ W, double* __restrict__ maps) {
  constexpr int C = 32 / R, GROUPS = BIG_THREADS / TR, CB = GROUPS * C;  // columns per workgroup
  const int J = Gm.J, N = Gm.N, c = blockIdx.x + 1, tid = threadIdx.x;      // (chunk 0 starts from the zero state: its map is never applied)
  const int rb = tid % TR, grp = tid / TR;
  const int r0 = rb * R, col0 = blockIdx.y * CB + grp * C;
  double A[R][C];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int k = 0; k < C; ++k) A[i][k] = (r0 + i == col0 + k) ? 1.0 : 0.0;
  const int s0 = c * Gm.L + 1, s1 = min(s0 + Gm.L, N);
  // the step's rows of the factor: forward n = s: decay phi[n-1], W[n-1], u[n-2];  backward n = N-1-s: phi[n], u[n], W[n+1]
  double pr[R], wr[R], ur[R], npr[R], nwr[R], nur[R];
  auto fetch = [&](int s) {
    const int n = BACKWARD ? N - 1 - s : s;
    const long op = (long)J * (BACKWARD ? n : n - 1), ow = (long)J * (BACKWARD ? n + 1 : n - 1), ou = (long)J * (BACKWARD ? n : n - 2);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int r = r0 + i;
      const bool ok = r < J && s < s1;
      npr[i] = ok ? phi[op + r] : 1.0;
      // the vector that multiplies the inner product (forward: W, backward: u) and the one inside it (forward: u, backward: W)
      nwr[i] = ok ? (BACKWARD ? u[op + r] : W[ow + r]) : 0.0;
      nur[i] = ok ? (BACKWARD ? W[ow + r] : u[ou + r]) : 0.0;
    }
  };
  fetch(s0);
  for (int s = s0; s < s1; ++s) {
#pragma unroll
    for (int i = 0; i < R; ++i) { pr[i] = npr[i]; wr[i] = nwr[i]; ur[i] = nur[i]; }
    fetch(s + 1);
    double t[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < R; ++i) acc = fma(ur[i], A[i][k], acc);
      t[k] = big_group_sum<TR>(acc);
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int k = 0; k < C; ++k) A[i][k] = pr[i] * fma(-wr[i], t[k], A[i][k]);
  }
  // column-major J x J per chunk: the walk's threads (one per row) read a column coalesced
  double* out = maps + (size_t)c * J * J;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int col = col0 + k;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int r = r0 + i;
      if (r < J && col < J) out[(size_t)col * J + r] = A[i][k];
    }
  }
}

// One workgroup per (chunk, right-hand side), thread = row.  REPLAY = false: from the zero state, the end state to `ends`;
// true: from starts[c], writing x (and the chunk's share of sum x^2 / D).  BACKWARD reads z = in / D on the fly.
template <bool REPLAY, bool BACKWARD, int NW>
__global__ void __launch_bounds__(64 * NW) bigsweep_chunk_kernel(BigGeom Gm, const double* __restrict__ phi, const double* __restrict__ u,
                                                                const double* __restrict__ W, const double* __restrict__ D,
                                                                const double* in, double* out /* never the same array as `in` */,
                                                                const double* __restrict__ starts, double* __restrict__ ends,
                                                                double* __restrict__ part) {
  __shared__ double xw[2][NW];
  const int J = Gm.J, N = Gm.N, c = blockIdx.x, rhs = blockIdx.y, row = threadIdx.x, lane = row & 63, wave = row >> 6;
  const bool have = row < J;
  const double* b = in + (size_t)rhs * N;
  double* x = out ? out + (size_t)rhs * N : nullptr;
  const size_t slot = ((size_t)rhs * Gm.nchunk + c) * J;
  double f = (REPLAY && have && c > 0) ? starts[slot + row] : 0.0;
  const int s0 = c * Gm.L + 1, s1 = min(s0 + Gm.L, N);
  double quad = 0.0;
  // the value that enters the first step: forward x_{s0-1} (= b_0 for chunk 0, else b - u . f of the START state),
  // backward the final x_{n+1} of sample n + 1 = N - s0 (= z_{N-1} for chunk 0)
  auto rows_sum = [&](double mine, int par) {  // sum over all rows, to every thread; one barrier (buffers alternate)
    const double ws = row_sum_all<1>(mine);
    if (NW == 1) return ws;
    if (lane == 0) xw[par][wave] = ws;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += xw[par][w];
    return tot;
  };
  double xprev;
  int par = 0;
  if (!BACKWARD) {
    const int n = s0 - 1;  // (sample before the chunk's first step)
    double dot = 0.0;
    if (n >= 1) { dot = rows_sum(have ? u[(size_t)J * (n - 1) + row] * f : 0.0, par); par ^= 1; }
    xprev = b[n] - dot;
    if (REPLAY && c == 0 && row == 0) { if (x) x[0] = xprev; }
    if (REPLAY && c == 0) quad = xprev * (xprev / D[0]);  // cholesky.h:347
  } else {
    const int n = N - s0;  // sample n + 1 of the chunk's first step n = N - 1 - s0
    double dot = 0.0;
    if (n <= N - 2) { dot = rows_sum(have ? W[(size_t)J * n + row] * f : 0.0, par); par ^= 1; }
    xprev = b[n] / D[n] - dot;  // :249, :256
    if (REPLAY && c == 0 && row == 0) x[N - 1] = xprev;
  }
  // the step's row of the factor and its right-hand side entry, fetched one step ahead (a load at the top of the step
  // would sit on the recurrence's critical path)
  double np_ = 1.0, nwv = 0.0, nov = 0.0, nbn = 0.0;
  auto fetch = [&](int s) {
    if (s >= s1) return;
    const int n = BACKWARD ? N - 1 - s : s;
    if (have) {
      if (!BACKWARD) { np_ = phi[(size_t)J * (n - 1) + row]; nwv = W[(size_t)J * (n - 1) + row]; nov = u[(size_t)J * (n - 1) + row]; }
      else { np_ = phi[(size_t)J * n + row]; nwv = u[(size_t)J * n + row]; nov = W[(size_t)J * n + row]; }
    }
    nbn = BACKWARD ? b[n] / D[n] : b[n];
  };
  fetch(s0);
  for (int s = s0; s < s1; ++s) {
    const int n = BACKWARD ? N - 1 - s : s;
    const double p = np_, wv = nwv, ov = nov, bn = nbn;
    fetch(s + 1);
    f = p * (f + wv * xprev);  // :350-352 / :253-255
    const bool last = s + 1 == s1;
    if (REPLAY || !last) {  // (the zero-start pass needs no output of its last step: only the end state)
      const double dot = rows_sum(ov * f, par);
      par ^= 1;
      const double xn = bn - dot;  // :353 / :256
      if (REPLAY) {
        if (row == 0 && x) x[n] = xn;
        if (!BACKWARD) quad += xn * xn / D[n];  // :355
      }
      xprev = xn;
    }
  }
  if (!REPLAY && have) ends[slot + row] = f;
  if (REPLAY && !BACKWARD && part && row == 0) part[(size_t)rhs * Gm.nchunk + c] = quad;
}

// F_0 = 0, F_{c+1} = A_c F_c + a_c: one workgroup of 1024 threads per right-hand side; thread (row r, slice q of SL): the
// columns q, q + SL, ... of its row (for a fixed column the threads of a slice read consecutive rows: coalesced)
template <int RPAD>
__global__ void __launch_bounds__(1024) bigsweep_walk_kernel(BigGeom Gm, const double* __restrict__ maps, const double* __restrict__ ends,
                                                             double* __restrict__ starts) {
  constexpr int SL = 1024 / RPAD;
  __shared__ double F[RPAD], partial[SL][RPAD];
  const int J = Gm.J, rhs = blockIdx.x, row = threadIdx.x % RPAD, q = threadIdx.x / RPAD;
  const bool have = row < J;
  if (q == 0) F[row] = 0.0;
  __syncthreads();
  for (int c = 0; c + 1 < Gm.nchunk; ++c) {
    const size_t slot = ((size_t)rhs * Gm.nchunk + c) * J;
    double acc = 0.0;
    if (c > 0 && have) {
      const double* A = maps + (size_t)c * J * J + row;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int col = q;
      for (; col + 3 * SL < J; col += 4 * SL) {
        a0 = fma(A[(size_t)col * J], F[col], a0);
        a1 = fma(A[(size_t)(col + SL) * J], F[col + SL], a1);
        a2 = fma(A[(size_t)(col + 2 * SL) * J], F[col + 2 * SL], a2);
        a3 = fma(A[(size_t)(col + 3 * SL) * J], F[col + 3 * SL], a3);
      }
      for (; col < J; col += SL) a0 = fma(A[(size_t)col * J], F[col], a0);
      acc = (a0 + a1) + (a2 + a3);
    }
    partial[q][row] = acc;
    __syncthreads();
    if (q == 0 && have) {
      double v = ends[slot + row];
#pragma unroll
      for (int k = 0; k < SL; ++k) v += partial[k][row];
      F[row] = v;
      starts[slot + J + row] = v;  // start state of chunk c + 1
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64) bigsweep_quad_kernel(int nchunk, const double* part, double* quad) {
  const int rhs = blockIdx.x, lane = threadIdx.x;
  double q = 0.0;
  for (int c = lane; c < nchunk; c += 64) q += part[(size_t)rhs * nchunk + c];
  q = row_sum_all<1>(q);
  if (lane == 0) quad[rhs] = q;
}

int big_waves(int J) { return J <= 128 ? 2 : (J <= 256 ? 4 : (J <= 512 ? 8 : 16)); }

}  // namespace

bool bigsweep_supported(int N, int J) { return J > 64 && J <= CLR_MAX_WIDTH_ANY && N >= 4096; }

// chunks: >= 128 steps each, at most 256; the maps of one direction stay under ~1 GB
// The walk reads nchunk maps of J^2 doubles through ONE compute unit (~100 GB/s), the two chunk passes cost ~2 N / nchunk
// steps of 1.5 .. 3 us: nchunk ~ sqrt(2 N t_step / t_map) (profiles/r06w_bigsweep.txt).
int bigsweep_chunks(int N, int J) {
  const double t_map = std::max(1.5e-6, 8.0 * J * (double)J / 1.0e11), t_step = J <= 128 ? 1.2e-6 : (J <= 512 ? 2.0e-6 : 3.0e-6);
  long nc = (long)sqrt(2.0 * N * t_step / t_map);
  nc = std::min<long>(nc, std::min<long>(512, (N - 1) / 64));
  const long cap = std::max<long>(2, (1L << 27) / ((long)J * J));
  nc = std::min(nc, cap);
  return (int)std::max<long>(2, nc);
}

size_t bigsweep_maps_doubles(int J, int nchunk) { return (size_t)nchunk * J * J; }
// ends | starts | partials
size_t bigsweep_workspace_doubles(int J, int nchunk, int nrhs) { return (size_t)nrhs * nchunk * (2 * (size_t)J + 1); }

void launch_bigsweep_maps(int N, int J, int nchunk, int L, int backward, const double* phi, const double* u, const double* W, double* maps, hipStream_t s) {
  if (nchunk < 2) return;
  BigGeom Gm{N, J, nchunk, L};
#define CLR_BIG_MAPS(R, TR)                                                                                                     \
  do {                                                                                                                          \
    constexpr int CB = (BIG_THREADS / TR) * (32 / R);                                                                           \
    const dim3 grid(nchunk - 1, (J + CB - 1) / CB);                                                                             \
    if (backward) hipLaunchKernelGGL((bigsweep_maps_kernel<R, TR, true>), grid, dim3(BIG_THREADS), 0, s, Gm, phi, u, W, maps);  \
    else hipLaunchKernelGGL((bigsweep_maps_kernel<R, TR, false>), grid, dim3(BIG_THREADS), 0, s, Gm, phi, u, W, maps);          \
  } while (0)
  if (J <= 128) CLR_BIG_MAPS(8, 16);
  else if (J <= 256) CLR_BIG_MAPS(8, 32);
  else if (J <= 512) CLR_BIG_MAPS(8, 64);
  else CLR_BIG_MAPS(16, 64);
#undef CLR_BIG_MAPS
}

// P.in -> P.out (forward: the undivided x, or null with P.quad; backward: in = the forward pass' output, out = K^-1 b)
void launch_bigsweep_scan(const SweepParams& P, const double* maps, double* workspace, hipStream_t s) {
  BigGeom Gm{P.N, P.J, P.nchunk, P.L};
  const size_t pc = (size_t)P.nrhs * P.nchunk;
  double* ends = workspace;
  double* starts = ends + pc * P.J;
  double* part = starts + pc * P.J;
  const int nw = big_waves(P.J);
  const dim3 grid(P.nchunk, P.nrhs), block(64 * nw);
#define CLR_BIG_CHUNK(REPLAY, NWV)                                                                                                           \
  do {                                                                                                                                        \
    if (P.backward) hipLaunchKernelGGL((bigsweep_chunk_kernel<REPLAY, true, NWV>), grid, block, 0, s, Gm, P.phi, P.u, P.W, P.D, P.in, P.out,   \
                                       (const double*)starts, ends, part);                                                                   \
    else hipLaunchKernelGGL((bigsweep_chunk_kernel<REPLAY, false, NWV>), grid, block, 0, s, Gm, P.phi, P.u, P.W, P.D, P.in, P.out,             \
                            (const double*)starts, ends, part);                                                                              \
  } while (0)
#define CLR_BIG_BOTH(REPLAY)                 \
  do {                                       \
    if (nw == 2) CLR_BIG_CHUNK(REPLAY, 2);   \
    else if (nw == 4) CLR_BIG_CHUNK(REPLAY, 4); \
    else if (nw == 8) CLR_BIG_CHUNK(REPLAY, 8); \
    else CLR_BIG_CHUNK(REPLAY, 16);          \
  } while (0)
  CLR_BIG_BOTH(false);
  if (P.J <= 128) hipLaunchKernelGGL((bigsweep_walk_kernel<128>), dim3(P.nrhs), dim3(1024), 0, s, Gm, maps, (const double*)ends, starts);
  else if (P.J <= 256) hipLaunchKernelGGL((bigsweep_walk_kernel<256>), dim3(P.nrhs), dim3(1024), 0, s, Gm, maps, (const double*)ends, starts);
  else if (P.J <= 512) hipLaunchKernelGGL((bigsweep_walk_kernel<512>), dim3(P.nrhs), dim3(1024), 0, s, Gm, maps, (const double*)ends, starts);
  else hipLaunchKernelGGL((bigsweep_walk_kernel<1024>), dim3(P.nrhs), dim3(1024), 0, s, Gm, maps, (const double*)ends, starts);
  CLR_BIG_BOTH(true);
  if (!P.backward && P.quad) hipLaunchKernelGGL(bigsweep_quad_kernel, dim3(P.nrhs), dim3(64), 0, s, P.nchunk, (const double*)part, P.quad);
#undef CLR_BIG_BOTH
#undef CLR_BIG_CHUNK
}

}  // namespace clr
