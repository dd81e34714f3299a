// celerite_amd/csrc/wsweep_kernels.hip -- dot_solve / solve on a stored factor of width <= 64 (and dot_L at any
// width) as chunked affine scans, one WAVE per chunk (single-solver API, series of N >= 2048; N >= 512 above width 8).
//
// The sweeps (cholesky.h:236-260, :343-357) are f <- p o (f + g x_prev) ; x = in - h . f with
// (p, g, h, in) = (phi, W, u, b) forward and (phi, u, W, x / D) backward: AFFINE on z = (f, x) in
// R^(J+1), so a chunk composes into z_end = A z_start + c.  At these widths A ((J+1)^2 <= 1089 doubles)
// does not fit a lane, but its COLUMNS evolve independently under the same per-step data:
//   summarize  one WAVE per chunk, one LANE per column of [A | c_1 .. c_nrhs]: every lane applies the
//              step to its own column (J + 1 doubles in registers); the step's p, g, h are wave-uniform
//              (scalar loads);
//   prefix     one wave per right-hand side walks the chunks: lane i owns z_i, z_j is handed round with
//              v_readlane, the chunk's matrix is read column-major (coalesced);
//   replay     one wave per (chunk, right-hand side): the reference recurrence from the known start,
//              lane = row, DPP reductions, factor rows prefetched KB steps ahead (as generic_kernels.hip).
// A sequential sweep costs 0.23 us per step (25-50 ms at N = 1e5, where the CPU needs 0.6-4 ms); measured here
// at N = 1e5 (profiles/r02y_sweeps_*): dot_solve 0.29 ms at width 8, 0.49 ms at width 32; solve twice that.
#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_options.h"
#include "clr_wide.h"

#include <stdlib.h>

#include <algorithm>

namespace clr {

namespace {

__device__ __forceinline__ double wsum(double v) { return row_sum<1>(v); }

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// problem blockIdx.z of a batched launch (SweepParams::batch): every per-problem pointer advanced
__device__ __forceinline__ SweepParams batch_view(SweepParams P) {
  const long b = blockIdx.z;
  if (b == 0) return P;
  P.phi += b * P.stride_phi; P.u += b * P.stride_phi; P.W += b * P.stride_W; P.D += b * P.stride_D;
  P.in += b * P.stride_in;
  if (P.out) P.out += b * P.stride_out;
  if (P.quad) P.quad += b * P.nrhs;
  P.elems += b * P.stride_ws; P.starts += b * P.stride_ws;
  if (P.part) P.part += b * P.stride_ws;
  if (P.run_starts) P.run_starts += b * P.stride_ws;
  return P;
}

// One wave per (chunk, 64 columns).  The step data (p, g, h: 3 J doubles per step) is the same for every
// lane: the wave copies KB steps at a time into LDS with coalesced vector loads (fetched one tile ahead,
// in registers) and every lane reads it back with uniform-address (broadcast) LDS reads.  (Through the
// scalar cache instead -- 96 doubles per step at width 32 against ~100 SGPRs -- a step cost 2.8 us.)
// The reads are software-pipelined by hand, GS rows ahead of the arithmetic, with scheduling barriers: left
// alone the compiler keeps ~3 reads in flight and a step pays the LDS latency a dozen times.
// The tile is padded so that the inner loops have no branch: rows J .. JP-1 are (0, 0, 0), steps past the
// chunk's end are (1, 0, 0) -- the identity on f -- and keep x through a select.
template <int JP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
wsweep_summarize_kernel(const SweepParams P0) {
  const SweepParams P = batch_view(P0);
  constexpr int KB = 8, ROUNDS = KB * JP / 64;
  constexpr int GS = (JP % 16 == 0) ? 16 : 8, NG = JP / GS;
  __shared__ double tile[KB][3][JP];
  const int J = P.J, K = J + 1, lane = threadIdx.x;
  const int c = blockIdx.x;
  const int id = blockIdx.y * 64 + lane;  // column of [A | c_0 .. c_{nrhs-1}]
  const bool live = id < K + P.nrhs;
  const bool affine = live && id >= K;
  const double* in = affine ? P.in + (long)(id - K) * P.N : nullptr;
  double f[JP], x = (live && id == J) ? 1.0 : 0.0;  // this lane's column: rows f_0 .. f_{J-1}, then x
#pragma unroll
  for (int i = 0; i < JP; ++i) f[i] = (live && i == id) ? 1.0 : 0.0;
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  const double* gp = P.backward ? P.u : P.W;
  const double* hp = P.backward ? P.W : P.u;
  double np[ROUNDS], ng[ROUNDS], nh[ROUNDS], nin[KB];
  auto fetch = [&](int sb) {
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const int r = lane + 64 * q, k = r / JP, i = r % JP, s = sb + k;
      const bool ok = s < s1 && i < J;
      const long col = (long)J * (P.backward ? P.N - 1 - s : s - 1);
      np[q] = ok ? P.phi[col + i] : (i < J ? 1.0 : 0.0);
      ng[q] = ok ? gp[col + i] : 0.0;
      nh[q] = ok ? hp[col + i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int s = sb + k, n = P.backward ? P.N - 1 - s : s;
      nin[k] = 0.0;
      if (affine && s < s1) nin[k] = P.backward ? in[n] / P.D[n] : in[n];
    }
  };
  fetch(s0);
  for (int sb = s0; sb < s1; sb += KB) {
    lds_fence();  // (the previous tile's reads have landed)
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const int r = lane + 64 * q, k = r / JP, i = r % JP;
      tile[k][0][i] = np[q];
      tile[k][1][i] = ng[q];
      tile[k][2][i] = nh[q];
    }
    double cin[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) cin[k] = nin[k];
    lds_fence();
    if (sb + KB < s1) fetch(sb + KB);
    double cp[GS], cg[GS], ch[GS];
#pragma unroll
    for (int i = 0; i < GS; ++i) { cp[i] = tile[0][0][i]; cg[i] = tile[0][1][i]; ch[i] = tile[0][2][i]; }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        double qp[GS], qg[GS], qh[GS];
        const int gn = (g + 1) % NG, kn = (g + 1 == NG) ? k + 1 : k;
        if (kn < KB) {
#pragma unroll
          for (int i = 0; i < GS; ++i) {
            qp[i] = tile[kn][0][gn * GS + i]; qg[i] = tile[kn][1][gn * GS + i]; qh[i] = tile[kn][2][gn * GS + i];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GS; ++i) {
          f[g * GS + i] = cp[i] * fma(cg[i], x, f[g * GS + i]);
          acc = fma(ch[i], f[g * GS + i], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kn < KB) {
#pragma unroll
          for (int i = 0; i < GS; ++i) { cp[i] = qp[i]; cg[i] = qg[i]; ch[i] = qh[i]; }
        }
      }
      x = (sb + k < s1) ? cin[k] - acc : x;
    }
  }
  if (!live) return;
  double* o = P.elems + ((long)c * (K + P.nrhs) + id) * K;  // column-major, K doubles per column
#pragma unroll
  for (int i = 0; i < JP; ++i)
    if (i < J) o[i] = f[i];
  o[J] = x;
}

// Two-level prefix (round 4): the maps of a RUN of consecutive chunks compose to one map of the same layout,
//     [P | a_1 .. a_nrhs]  <-  M_c [P | a] + [0 | a_c] ,   from [I | 0],
// so the walk over the chunks (0.34-0.6 us each, a third of a sweep) becomes: compose the runs side by side, walk the
// run maps, walk every run's own chunks from the state found for its start.  One wave per (run, block of CB columns);
// lane = row, the block's columns of the running product in registers, handed round through LDS.
constexpr int WS_CB = 8;
__global__ void __launch_bounds__(64) wsweep_compose_kernel(const SweepParams P0, int R, double* run_elems0) {
  const SweepParams P = batch_view(P0);
  double* run_elems = run_elems0 + (long)blockIdx.z * P0.stride_ws;
  __shared__ double X[65][WS_CB];
  const int J = P.J, K = J + 1, lane = threadIdx.x, run = blockIdx.x, col0 = blockIdx.y * WS_CB;
  const int ncol = K + P.nrhs;
  const int c0 = run * R, c1 = min(c0 + R, P.nchunk);
  // (width 64: K = 65 rows -- lane 0 also owns row 64)
  const bool have = lane < K, have2 = lane == 0 && K > 64;
  double acc[WS_CB], acc2[WS_CB];
#pragma unroll
  for (int j = 0; j < WS_CB; ++j) {
    acc[j] = (have && col0 + j == lane) ? 1.0 : 0.0;
    acc2[j] = (have2 && col0 + j == K - 1) ? 1.0 : 0.0;
  }
  const int lrow = min(lane, K - 1);
  for (int c = c0; c < c1; ++c) {
    const double* M = P.elems + (long)c * ncol * K;
    if (have) {
#pragma unroll
      for (int j = 0; j < WS_CB; ++j) X[lane][j] = acc[j];
    }
    if (have2) {
#pragma unroll
      for (int j = 0; j < WS_CB; ++j) X[K - 1][j] = acc2[j];
    }
    lds_fence();
    double nacc[WS_CB], nacc2[WS_CB];
#pragma unroll
    for (int j = 0; j < WS_CB; ++j) {
      const int col = col0 + j;
      const bool aff = col >= K && col < ncol;
      nacc[j] = aff ? M[(long)col * K + lrow] : 0.0;
      nacc2[j] = (aff && have2) ? M[(long)col * K + K - 1] : 0.0;
    }
    for (int k = 0; k < K; ++k) {
      const double m = M[(long)k * K + lrow];
      const double m2 = have2 ? M[(long)k * K + K - 1] : 0.0;
#pragma unroll
      for (int j = 0; j < WS_CB; ++j) {
        const double x = X[k][j];
        nacc[j] = fma(m, x, nacc[j]);
        nacc2[j] = fma(m2, x, nacc2[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < WS_CB; ++j) { acc[j] = nacc[j]; acc2[j] = nacc2[j]; }
    lds_fence();
  }
  double* O = run_elems + (long)run * ncol * K;
#pragma unroll
  for (int j = 0; j < WS_CB; ++j) {
    const int col = col0 + j;
    if (col < ncol) {
      if (have) O[(long)col * K + lane] = acc[j];
      if (have2) O[(long)col * K + K - 1] = acc2[j];
    }
  }
}

// One workgroup of NW waves per right-hand side.  Lane i (< K) owns row i of a chunk's map; wave w owns the
// chunks c = w (mod NW) and fetches its next one right after using the current, so a fetch has NW - 1 chunk
// times to land; z travels from chunk to chunk through LDS (ping-pong, one barrier per chunk) and is read
// back with broadcast reads.
template <int JP, int NW>
__global__ void __launch_bounds__(64 * NW) wsweep_prefix_kernel(const SweepParams P0) {
  const SweepParams P = batch_view(P0);
  constexpr int H = (JP + 2) / 2;  // z is read back in two halves of H (columns 0 .. JP+1, the last one padding)
  constexpr bool BIG = JP >= 64;   // width 64: K = 65 rows for 64 lanes -- lane 0 also owns row 64 (the x row)
  __shared__ double zbuf[2][2 * H];
  const int J = P.J, K = J + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rhs = blockIdx.x;
  // (two-level prefix: this workgroup walks the chunks of run blockIdx.y from the state the level above found for it)
  const int c_lo = P.run_len ? (int)blockIdx.y * P.run_len : 0;
  const int c_hi = P.run_len ? min(c_lo + P.run_len, P.nchunk) : P.nchunk;
  const double* z_lo = P.run_starts ? P.run_starts + ((long)rhs * gridDim.y + blockIdx.y) * K : nullptr;
  const bool have = lane < K;
  const bool have2 = BIG && lane == 0 && K > 64;
  const double* in = P.in + (long)rhs * P.N;
  // (no aliasing between the maps and the starts: the fetch must not wait for the store of a start)
  const double* __restrict__ elems = P.elems;
  double* __restrict__ starts = P.starts;
  if (wave == 0) {
    // the sweep's first sample: x_0 = b_0 (cholesky.h:238) / x_{N-1} / D_{N-1} (:249,251)
    for (int i = lane; i < 2 * H; i += 64) {
      double z = 0.0;
      if (z_lo) z = i < K ? z_lo[i] : 0.0;
      else if (i == J) z = P.backward ? in[P.N - 1] / P.D[P.N - 1] : in[0];
      zbuf[0][i] = z;
      zbuf[1][i] = 0.0;
    }
  }
  double m[2 * H], aff = 0.0;
  // BIG: row K-1 of the map spread over the lanes (lane j holds its entry in column j), the 65th column and the
  // affine part wave-uniform; the row's product with z is a wave reduction
  double m2 = 0.0, m2last = 0.0, aff2 = 0.0;
  const int lrow = min(lane, K - 1);
  // every address in range and no select on the loaded values (a select would wait for them right here):
  // the padding columns j >= K re-read column K-1 and meet z_j = 0; lanes >= K compute a value nobody stores
  auto fetch = [&](int c) {
    const double* M = elems + (long)c * (K + P.nrhs) * K;
#pragma unroll
    for (int j = 0; j < 2 * H; ++j) m[j] = M[(long)min(j, K - 1) * K + lrow];
    aff = M[(long)(K + rhs) * K + lrow];
    if (BIG) {
      m2 = M[(long)min(lane, K - 1) * K + (K - 1)];
      m2last = M[(long)(K - 1) * K + (K - 1)];
      aff2 = M[(long)(K + rhs) * K + (K - 1)];
    }
  };
  if (c_lo + wave < c_hi) fetch(c_lo + wave);
  __syncthreads();
  for (int c = c_lo; c < c_hi; ++c) {
    if ((c - c_lo) % NW == wave) {
      const double* zc = zbuf[(c - c_lo) & 1];
      double z0[H], z1[H];
#pragma unroll
      for (int j = 0; j < H; ++j) z0[j] = zc[j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < H; ++j) z1[j] = zc[H + j];
      const double zmine = zc[min(lane, 2 * H - 1)];
      const double zlast = zc[K - 1];
      double a0 = aff, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // (z_j = 0 for the padding columns j >= K)
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (j % 2 == 0) a0 = fma(m[j], z0[j], a0); else a1 = fma(m[j], z0[j], a1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (j % 2 == 0) a2 = fma(m[H + j], z1[j], a2); else a3 = fma(m[H + j], z1[j], a3);
      }
      double b0 = 0.0;
      if (BIG && K > 64) b0 = wsum(m2 * zmine) + fma(m2last, zlast, aff2);  // (lane j: column j < 64; then column 64)
      if (have) zbuf[(c - c_lo + 1) & 1][lane] = (a0 + a1) + (a2 + a3);
      if (have2) zbuf[(c - c_lo + 1) & 1][K - 1] = b0;
      // (the store after the arithmetic: issued before it, the wait for this chunk's map -- vmcnt counts
      // loads and stores in order -- would also wait for the store to be acknowledged)
      if (have) starts[((long)rhs * P.nchunk + c) * K + lane] = zmine;
      if (have2) starts[((long)rhs * P.nchunk + c) * K + K - 1] = zlast;
      if (c + NW < c_hi) fetch(c + NW);
    }
    // LDS traffic only: a plain __syncthreads() would also wait for the fetch just issued (vmcnt(0))
    lds_fence();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

// one wave per (chunk, right-hand side); lane = row of the state
__global__ void __launch_bounds__(64) wsweep_replay_kernel(const SweepParams P0) {
  const SweepParams P = batch_view(P0);
  constexpr int KB = 8;
  const int J = P.J, K = J + 1, lane = threadIdx.x, c = blockIdx.x, rhs = blockIdx.y;
  const bool have = lane < J;
  const double* in = P.in + (long)rhs * P.N;
  double* out = P.out ? P.out + (long)rhs * P.N : nullptr;
  const double* st0 = P.starts + ((long)rhs * P.nchunk + c) * K;
  double f = have ? st0[lane] : 0.0, x = st0[J], quad = 0.0;
  if (c == 0) {  // the first sample of the sweep belongs to chunk 0
    const int n0 = P.backward ? P.N - 1 : 0;
    if (out && lane == 0) out[n0] = x;
    quad = x * (x / P.D[n0]);  // cholesky.h:347
  }
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  const double* gp = P.backward ? P.u : P.W;
  const double* hp = P.backward ? P.W : P.u;
  double np[KB], ng[KB], nh[KB], nin = 0.0, nd = 1.0;
  auto fetch = [&](int sb) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int s = sb + k;
      const int n = P.backward ? P.N - 1 - s : s;
      const long col = (long)J * (P.backward ? n : s - 1);
      const bool ok = s < s1 && have;
      np[k] = ok ? P.phi[col + lane] : 0.0;
      ng[k] = ok ? gp[col + lane] : 0.0;
      nh[k] = ok ? hp[col + lane] : 0.0;
    }
    const int s = sb + lane;
    const int n = P.backward ? P.N - 1 - s : s;
    const bool ok = lane < KB && s < s1;
    nd = ok ? P.D[n] : 1.0;
    nin = ok ? (P.backward ? in[n] / nd : in[n]) : 0.0;
  };
  fetch(s0);
  for (int sb = s0; sb < s1; sb += KB) {
    double cp[KB], cg[KB], ch[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { cp[k] = np[k]; cg[k] = ng[k]; ch[k] = nh[k]; }
    const double cin = nin, cd = nd;
    if (sb + KB < s1) fetch(sb + KB);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int s = sb + k;
      if (s < s1) {
        f = cp[k] * fma(cg[k], x, f);            // cholesky.h:243-246 / :255-258 / :350-354
        x = lane_value(cin, k) - wsum(ch[k] * f);
        const int n = P.backward ? P.N - 1 - s : s;
        if (out && lane == 0) out[n] = x;
        quad += x * x / lane_value(cd, k);       // :356
      }
    }
  }
  if (P.part && lane == 0) P.part[(long)rhs * P.nchunk + c] = quad;
}

// dot_solve: chunk partials, one wave per right-hand side (lane-strided sums, then a fixed reduction tree)
__global__ void __launch_bounds__(64) wsweep_finalize_kernel(const SweepParams P0) {
  const SweepParams P = batch_view(P0);
  const int rhs = blockIdx.x, lane = threadIdx.x;
  double q = 0.0;
  for (int c = lane; c < P.nchunk; c += 64) q += P.part[(long)rhs * P.nchunk + c];
  q = wsum(q);
  if (lane == 0) P.quad[rhs] = q;
}

// dot_L (cholesky.h:409-431) at any width: f_j <- phi_j (f_j + W_j sqrt(D_{n-1}) z_{n-1}) is DIAGONAL in j,
// so a chunk is (a_j, f_j) per row; NW waves per (chunk, right-hand side), thread = row (round 6: widths above 64 --
// the replay's sum over the rows then goes through LDS, one workgroup barrier per tile of KB steps).
template <bool REPLAY, int NW>
__global__ void __launch_bounds__(64 * NW) wdotl_kernel(const SweepParams P0) {
  const SweepParams P = batch_view(P0);  // (clr_batch_dot_L on the wide plans: grid.z = problem)
  constexpr int KB = 8;
  __shared__ double xw[2][NW][KB];
  const int J = P.J, row = threadIdx.x, lane = row & 63, wave = row >> 6, c = blockIdx.x, rhs = blockIdx.y;
  const bool have = row < J;
  const double* z = P.in + (long)rhs * P.N;
  double* y = P.out + (long)rhs * P.N;
  const long slot = (long)rhs * P.nchunk + c;
  double a = 1.0, f = (REPLAY && have) ? P.starts[slot * J + row] : 0.0;
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  if (REPLAY && c == 0 && row == 0) y[0] = sqrt(P.D[0]) * z[0];  // :421-422
  double np[KB], nw[KB], nu[KB], ntz = 0.0;
  auto fetch = [&](int sb) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int n = sb + k;
      const long col = (long)J * (n - 1);
      const bool ok = n < s1 && have;
      np[k] = ok ? P.phi[col + row] : 0.0;
      nw[k] = ok ? P.W[col + row] : 0.0;
      nu[k] = (REPLAY && ok) ? P.u[col + row] : 0.0;
    }
    const int n = sb - 1 + lane;  // lanes 0..KB (of every wave): sqrt(D_n) z_n for n = sb-1 .. sb+KB-1
    ntz = (lane <= KB && n < s1) ? sqrt(P.D[n]) * z[n] : 0.0;
  };
  fetch(s0);
  int par = 0;
  for (int sb = s0; sb < s1; sb += KB, par ^= 1) {
    double cp[KB], cw[KB], cu[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { cp[k] = np[k]; cw[k] = nw[k]; cu[k] = nu[k]; }
    const double ctz = ntz;
    const double ctz_up = (REPLAY && NW > 1) ? __shfl(ctz, (lane + 1) & 63, 64) : 0.0;  // lane k: sqrt(D_n) z_n of step sb + k
    if (sb + KB < s1) fetch(sb + KB);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int n = sb + k;
      if (n < s1) {
        f = cp[k] * (f + cw[k] * lane_value(ctz, k));  // :424-425
        if (!REPLAY) a *= cp[k];
        if (REPLAY) {
          const double rows = wsum(cu[k] * f);
          if (NW == 1) {
            const double v = lane_value(ctz, k + 1) + rows;  // :426
            if (lane == 0) y[n] = v;
          } else if (lane == 0) xw[par][wave][k] = rows;
        }
      }
    }
    if (REPLAY && NW > 1) {  // the tile's sums over the waves (the buffer of the tile before last is free again: one barrier)
      __syncthreads();
      if (wave == 0 && lane < KB && sb + lane < s1) {
        double v = ctz_up;
        for (int w = 0; w < NW; ++w) v += xw[par][w][lane];
        y[sb + lane] = v;
      }
    }
  }
  if (!REPLAY && have) {
    P.elems[slot * 2 * J + row] = a;
    P.elems[slot * 2 * J + J + row] = f;
  }
}

__global__ void __launch_bounds__(1024) wdotl_prefix_kernel(const SweepParams P0) {  // (thread = row: 64 ... 1024 threads)
  const SweepParams P = batch_view(P0);
  constexpr int KB = 16;  // chunks fetched ahead
  const int J = P.J, lane = threadIdx.x, rhs = blockIdx.x;
  if (lane >= J) return;
  double f = 0.0, na[KB], ne[KB];
  auto fetch = [&](int cb) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const long slot = (long)rhs * P.nchunk + cb + k;
      const bool ok = cb + k < P.nchunk;
      na[k] = ok ? P.elems[slot * 2 * J + lane] : 1.0;
      ne[k] = ok ? P.elems[slot * 2 * J + J + lane] : 0.0;
    }
  };
  fetch(0);
  for (int cb = 0; cb < P.nchunk; cb += KB) {
    double ca[KB], ce[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { ca[k] = na[k]; ce[k] = ne[k]; }
    if (cb + KB < P.nchunk) fetch(cb + KB);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (cb + k < P.nchunk) {
        P.starts[((long)rhs * P.nchunk + cb + k) * J + lane] = f;
        f = fma(ca[k], f, ce[k]);
      }
    }
  }
}

// dot (cholesky.h:533-560): y = K z from phi, u, v (all J x N) and the diagonal dg -- two more diagonal
// recurrences, PASS 0 the upper triangle walking n down (:536-547), PASS 1 the lower one walking n up (:549-559),
// each as summarize / prefix (wdotl_prefix_kernel) / replay over chunks of the step index s = 1 .. N-1.
template <bool REPLAY, int PASS, int NW>
__global__ void __launch_bounds__(64 * NW) wdot_kernel(const SweepParams P, const double* __restrict__ v,
                                                       const double* __restrict__ dg) {
  constexpr int KB = 8;
  __shared__ double xw[2][NW][KB];
  const int J = P.J, N = P.N, row = threadIdx.x, lane = row & 63, wave = row >> 6, c = blockIdx.x, rhs = blockIdx.y;
  const bool have = row < J;
  const double* z = P.in + (long)rhs * N;
  double* y = P.out + (long)rhs * N;
  const long slot = (long)rhs * P.nchunk + c;
  double a = 1.0, f = (REPLAY && have) ? P.starts[slot * J + row] : 0.0;
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, N);
  if (REPLAY && PASS == 0 && c == 0 && row == 0) y[N - 1] = dg[N - 1] * z[N - 1];  // :535
  const double* wp = PASS == 0 ? P.u : v;   // weight of the incoming z
  const double* op = PASS == 0 ? v : P.u;   // weight of f in the output
  double np[KB], nw[KB], no[KB], nzin = 0.0, nbase = 0.0;
  auto fetch = [&](int sb) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int s = sb + k;
      const int n = PASS == 0 ? N - 1 - s : s;
      const long base = (long)J * (PASS == 0 ? n : n - 1);
      const bool ok = s < s1 && have;
      np[k] = ok ? P.phi[base + row] : 0.0;
      nw[k] = ok ? wp[base + row] : 0.0;
      no[k] = (REPLAY && ok) ? op[base + row] : 0.0;
    }
    const int s = sb + lane;  // lanes 0 .. KB-1 (of every wave): the step's scalars
    const int n = PASS == 0 ? N - 1 - s : s;
    const bool ok = lane < KB && s < s1;
    nzin = ok ? z[PASS == 0 ? n + 1 : n - 1] : 0.0;
    nbase = (REPLAY && ok) ? (PASS == 0 ? dg[n] * z[n] : y[n]) : 0.0;  // (PASS 1 adds to PASS 0's result)
  };
  fetch(s0);
  int par = 0;
  for (int sb = s0; sb < s1; sb += KB, par ^= 1) {
    double cp[KB], cw[KB], co[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { cp[k] = np[k]; cw[k] = nw[k]; co[k] = no[k]; }
    const double czin = nzin, cbase = nbase;
    if (sb + KB < s1) fetch(sb + KB);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int s = sb + k;
      if (s < s1) {
        f = cp[k] * (f + cw[k] * lane_value(czin, k));  // :540-542 / :552-554
        if (!REPLAY) a *= cp[k];
        if (REPLAY) {
          const double rows = wsum(co[k] * f);
          if (NW == 1) {
            const double val = lane_value(cbase, k) + rows;  // :543-545 / :555-557
            if (lane == 0) y[PASS == 0 ? N - 1 - s : s] = val;
          } else if (lane == 0) xw[par][wave][k] = rows;
        }
      }
    }
    if (REPLAY && NW > 1) {  // the tile's sums over the waves
      __syncthreads();
      if (wave == 0 && lane < KB && sb + lane < s1) {
        double val = cbase;  // (lane k holds step sb + k's base value)
        for (int w = 0; w < NW; ++w) val += xw[par][w][lane];
        const int s = sb + lane;
        y[PASS == 0 ? N - 1 - s : s] = val;
      }
    }
  }
  if (!REPLAY && have) {
    P.elems[slot * 2 * J + row] = a;
    P.elems[slot * 2 * J + J + row] = f;
  }
}

}  // namespace

// workspace: nrhs * nchunk * 3 J doubles; P.phi, P.u = phi, u of `dot`'s own setup (J x N), P.in = z, P.out = y
// waves per workgroup of the diagonal scans: thread = row
static int wdot_waves(int J) { return J <= 64 ? 1 : (J <= 128 ? 2 : (J <= 256 ? 4 : (J <= 512 ? 8 : 16))); }
#define CLR_WDOT_DISPATCH(NWV, CALL) \
  do {                               \
    switch (NWV) {                   \
      case 1: { constexpr int NW = 1; CALL; break; }   \
      case 2: { constexpr int NW = 2; CALL; break; }   \
      case 4: { constexpr int NW = 4; CALL; break; }   \
      case 8: { constexpr int NW = 8; CALL; break; }   \
      default: { constexpr int NW = 16; CALL; break; } \
    }                                \
  } while (0)

void launch_wdot_scan(SweepParams P, const double* v, const double* dg, double* workspace, hipStream_t s) {
  const size_t pc = (size_t)P.nrhs * P.nchunk;
  P.elems = workspace;
  P.starts = P.elems + pc * 2 * P.J;
  const dim3 grid(P.nchunk, P.nrhs);
  const int nwv = wdot_waves(P.J);
  const dim3 block(64 * nwv);
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdot_kernel<false, 0, NW>), grid, block, 0, s, P, v, dg));
  hipLaunchKernelGGL(wdotl_prefix_kernel, dim3(P.nrhs), block, 0, s, P);
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdot_kernel<true, 0, NW>), grid, block, 0, s, P, v, dg));
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdot_kernel<false, 1, NW>), grid, block, 0, s, P, v, dg));
  hipLaunchKernelGGL(wdotl_prefix_kernel, dim3(P.nrhs), block, 0, s, P);
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdot_kernel<true, 1, NW>), grid, block, 0, s, P, v, dg));
}

// long series at any width; above width 128 (no sequential kernel there) every series of two samples or more
bool wdotl_scan_supported(int N, int J) { return J >= 1 && J <= CLR_MAX_WIDTH_ANY && (N >= 2048 || (J > CLR_MAX_WIDTH && N >= 2)); }
int wdotl_chunks(int N) { return std::max(2, std::min(512, (N - 1) / 128)); }

// workspace: nrhs * nchunk * 3 J doubles (per problem of a batched launch: P.stride_ws apart)
void launch_wdotl_scan(SweepParams P, double* workspace, hipStream_t s) {
  const size_t pc = (size_t)P.nrhs * P.nchunk;
  P.elems = workspace;
  P.starts = P.elems + pc * 2 * P.J;
  const unsigned nb = P.batch > 1 ? (unsigned)P.batch : 1u;
  const dim3 grid(P.nchunk, P.nrhs, nb);
  const int nwv = wdot_waves(P.J);
  const dim3 block(64 * nwv);
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdotl_kernel<false, NW>), grid, block, 0, s, P));
  hipLaunchKernelGGL(wdotl_prefix_kernel, dim3(P.nrhs, 1, nb), block, 0, s, P);
  CLR_WDOT_DISPATCH(nwv, hipLaunchKernelGGL((wdotl_kernel<true, NW>), grid, block, 0, s, P));
}

// (measured at N = 1e5, width 8: 0.29 ms against 1.35 ms for the lane-per-chunk scan of sweep_kernels.hip, which
// keeps the series of 256 <= N < 2048)
// round 3: widths 33..64 too (summarize: a lane's column is up to 65 doubles, one wave per SIMD; prefix: at width 64 the
// chunk map has 65 rows, lane 0 owns two), and shorter series at widths above 8 (the lane-per-chunk scan of
// sweep_kernels.hip stops at width 8; a sequential sweep costs 0.23 us per step)
bool wsweep_scan_supported(int N, int J) { return J >= 1 && J <= 64 && (N >= 2048 || (J > 8 && N >= 512)); }

// runs of the two-level prefix: 0 = one walk (widths above 32: a composition costs (J + 1)^3)
static int wsweep_run_len(int J, int nchunk) {
  if (const char* e = clr::option("CLR_WSWEEP_RUN")) return atoi(e);  // (tools/gpu_wsweep_chunks.py)
  if (J > 32 || nchunk < 128) return 0;
  return 8;  // (profiles/r04z_wsweep_two_level.txt: runs of 8 beat 12, 16, 32 at every width <= 32)
}

// the sequential prefix costs 0.3-0.6 us per chunk, the parallel phases 0.4-0.8 us per step
int wsweep_chunks(int N, int J) {
  long nc = (long)(1.0 * sqrt((double)N));
  // two-level prefix (widths <= 32, long series): one round of 1024 chunk waves -- the walk no longer sets the chunk count
  // (N = 1e5: dot_solve 0.31 -> 0.18 ms at width 8, 0.44 -> 0.29 at width 32; 1536 chunks are slower again)
  if (J <= 32 && N >= 16384) nc = 1024;
  if (const char* e = clr::option("CLR_WSWEEP_CHUNKS")) nc = atol(e);
  if (nc < 2) nc = 2;
  const long maxc = std::max<long>(1, (N - 1) / 64);
  if (nc > maxc) nc = maxc;
  return (int)nc;
}

size_t wsweep_workspace_doubles(int J, int nchunk, int nrhs) {
  const size_t K = (size_t)J + 1;
  const int R = wsweep_run_len(J, nchunk);
  const size_t nrun = R > 0 ? (size_t)((nchunk + R - 1) / R) : 0;
  return (size_t)nchunk * (K + nrhs) * K + (size_t)nrhs * nchunk * K + (size_t)nrhs * nchunk +
         nrun * (K + nrhs) * K + (size_t)nrhs * nrun * K;
}

void launch_wsweep_scan(SweepParams P, double* workspace, hipStream_t s) {
  const size_t K = (size_t)P.J + 1;
  P.elems = workspace;
  P.starts = P.elems + (size_t)P.nchunk * (K + P.nrhs) * K;
  P.part = P.starts + (size_t)P.nrhs * P.nchunk * K;
  const unsigned nb = P.batch > 1 ? (unsigned)P.batch : 1u;
  const dim3 gsum(P.nchunk, (unsigned)((K + P.nrhs + 63) / 64), nb);
  if (P.J <= 8) hipLaunchKernelGGL((wsweep_summarize_kernel<8>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 16) hipLaunchKernelGGL((wsweep_summarize_kernel<16>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 24) hipLaunchKernelGGL((wsweep_summarize_kernel<24>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 32) hipLaunchKernelGGL((wsweep_summarize_kernel<32>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 40) hipLaunchKernelGGL((wsweep_summarize_kernel<40>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 48) hipLaunchKernelGGL((wsweep_summarize_kernel<48>), gsum, dim3(64), 0, s, P);
  else if (P.J <= 56) hipLaunchKernelGGL((wsweep_summarize_kernel<56>), gsum, dim3(64), 0, s, P);
  else hipLaunchKernelGGL((wsweep_summarize_kernel<64>), gsum, dim3(64), 0, s, P);
  auto prefix = [&](const SweepParams& Q, int nrun) {
    const dim3 grid(Q.nrhs, nrun, nb);
    if (Q.J <= 8) hipLaunchKernelGGL((wsweep_prefix_kernel<8, 16>), grid, dim3(1024), 0, s, Q);
    else if (Q.J <= 16) hipLaunchKernelGGL((wsweep_prefix_kernel<16, 16>), grid, dim3(1024), 0, s, Q);
    else if (Q.J <= 24) hipLaunchKernelGGL((wsweep_prefix_kernel<24, 8>), grid, dim3(512), 0, s, Q);
    else if (Q.J <= 32) hipLaunchKernelGGL((wsweep_prefix_kernel<32, 8>), grid, dim3(512), 0, s, Q);
    else if (Q.J <= 48) hipLaunchKernelGGL((wsweep_prefix_kernel<48, 4>), grid, dim3(256), 0, s, Q);
    else if (Q.J <= 62) hipLaunchKernelGGL((wsweep_prefix_kernel<62, 4>), grid, dim3(256), 0, s, Q);
    else hipLaunchKernelGGL((wsweep_prefix_kernel<64, 4>), grid, dim3(256), 0, s, Q);
  };
  const int R = wsweep_run_len(P.J, P.nchunk);
  if (R > 0 && P.nchunk > R) {
    // two levels: the runs' maps composed side by side, the run maps walked, every run walked from its start state
    const int nrun = (P.nchunk + R - 1) / R;
    double* run_elems = P.part + (size_t)P.nrhs * P.nchunk;
    double* run_starts = run_elems + (size_t)nrun * (K + P.nrhs) * K;
    hipLaunchKernelGGL(wsweep_compose_kernel, dim3(nrun, (unsigned)((K + P.nrhs + WS_CB - 1) / WS_CB), nb), dim3(64), 0, s, P, R, run_elems);
    SweepParams T = P;
    T.elems = run_elems; T.nchunk = nrun; T.starts = run_starts; T.run_len = 0; T.run_starts = nullptr;
    prefix(T, 1);
    SweepParams F = P;
    F.run_len = R; F.run_starts = run_starts;
    prefix(F, nrun);
  } else {
    P.run_len = 0; P.run_starts = nullptr;
    prefix(P, 1);
  }
  hipLaunchKernelGGL(wsweep_replay_kernel, dim3(P.nchunk, P.nrhs, nb), dim3(64), 0, s, P);
  if (P.quad) hipLaunchKernelGGL(wsweep_finalize_kernel, dim3(P.nrhs, 1, nb), dim3(64), 0, s, P);
}

}  // namespace clr
