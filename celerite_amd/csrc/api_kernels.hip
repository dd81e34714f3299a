// celerite_amd/csrc/api_kernels.hip -- the small kernels a plan launches itself: the per-problem finalize, the
// row-major <-> chunk-interleaved relayouts of the series, the de-interleave of a materialised factor; and the table of
// the per-width kernel launchers (batch_w*.hip).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/celerite_hip.h"
#include "clr_batch_kernels.h"
#include "clr_wide.h"

namespace clr {
const BatchLaunchers* batch_launchers_w1(int, int);
const BatchLaunchers* batch_launchers_w2(int, int);
const BatchLaunchers* batch_launchers_w3(int, int);
const BatchLaunchers* batch_launchers_w4(int, int);
const BatchLaunchers* batch_launchers_w5(int, int);
const BatchLaunchers* batch_launchers_w6(int, int);
const BatchLaunchers* batch_launchers_w7(int, int);
const BatchLaunchers* batch_launchers_w8(int, int);

const BatchLaunchers* find_batch_launchers(int JR, int JC) {
  switch (JR + 2 * JC) {
    case 1: return batch_launchers_w1(JR, JC);
    case 2: return batch_launchers_w2(JR, JC);
    case 3: return batch_launchers_w3(JR, JC);
    case 4: return batch_launchers_w4(JR, JC);
    case 5: return batch_launchers_w5(JR, JC);
    case 6: return batch_launchers_w6(JR, JC);
    case 7: return batch_launchers_w7(JR, JC);
    case 8: return batch_launchers_w8(JR, JC);
    default: return nullptr;
  }
}

// One wave per problem: lane l sums chunks l, l + 64, ... in order, then a fixed butterfly -- the same tree whatever the
// batch size or sharding, so results stay bit-identical across shard counts.  (One thread per problem walking all
// chunks took 40 us at 125 chunks: a fifth of BASELINE config 1's step.)
__global__ void __launch_bounds__(64) finalize_kernel(const BatchParams P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (P.only_pending && P.need_scan[b] == 0) return;  // (settled and written by the warm path)
  if (P.defer_level1 && !P.force_exact && P.need_exact[b] == 1) {
    // ill-conditioned but not flagged: the checked replay was deferred -- the host re-plans this problem with many
    // short chunks before results are handed out (api_batch.hip: rescue_run); until then its status says so
    if (lane == 0) {
      P.out_status[b] = CLR_PENDING_STATUS;
      P.out_ll[b] = NAN; P.out_logdet[b] = NAN; P.out_quad[b] = NAN;
    }
    return;
  }
  // replay-free sums unless the problem was marked for (or the run forces) the exact replay
  const bool exact = P.force_exact || P.need_exact[b] != 0;
  const double* part = exact ? P.partx : P.part;
  const int* flags = exact ? P.flagsx : P.flags;
  double ld = 0.0, qd = 0.0;
  int bad = 0;
  for (int c = lane; c < P.nchunk; c += 64) {
    ld += part[((long)b * P.nchunk + c) * 2 + 0];
    qd += part[((long)b * P.nchunk + c) * 2 + 1];
    bad |= flags[(long)b * P.nchunk + c];
  }
  ld = row_sum<1>(ld);
  qd = row_sum<1>(qd);
  bad = __any(bad) ? 1 : 0;
  if (lane != 0) return;
  if (bad) {  // celerite::linalg_exception (cholesky.h:176); quiet => -inf (celerite.py:205-208)
    P.out_status[b] = CLR_NOT_POSITIVE_DEFINITE;
    P.out_ll[b] = -INFINITY;
    P.out_logdet[b] = NAN;
    P.out_quad[b] = NAN;
    return;
  }
  P.out_status[b] = CLR_OK;
  P.out_logdet[b] = ld;
  P.out_quad[b] = qd;
  P.out_ll[b] = combine_loglike(ld, qd, P.N);
}

void launch_finalize(const BatchParams& P, hipStream_t s) {
  hipLaunchKernelGGL(finalize_kernel, dim3(P.B), dim3(64), 0, s, P);
}

// Tiled transpose through LDS: reads coalesced along i (the time axis), writes
// coalesced along the chunk axis.  Pure data movement: 8 B in + 8 B out per sample.
// Cells past the end of the series (the tail of the last chunk) are filled so that a reader may treat them as
// ordinary samples: pad_kind 1 repeats the series' last value (t: dx = 0), 2 writes 1e300 (the diagonal: 1 / D ~ 0),
// 0 writes zeros (y).  The lazy role-split summarize reads them unguarded (clr_split_kernels.h); every other reader
// masks them and never sees the values.
__global__ void __launch_bounds__(256) relayout_kernel(const double* __restrict__ src,
                                                       long src_stride, double* __restrict__ dst,
                                                       long dst_stride, int N, int L, int nchunk, int pad_kind) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z, i0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const double* in = src + (long)b * src_stride;
  double* out = dst + (long)b * dst_stride;
  const double pad = pad_kind == 1 ? in[N - 1] : (pad_kind == 2 ? 1e300 : 0.0);
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int c = c0 + r, i = i0 + threadIdx.x;
    const long n = (long)c * L + i;
    tile[r][threadIdx.x] = (c < nchunk && i < L && n < N) ? in[n] : pad;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = i0 + r, c = c0 + threadIdx.x;
    if (i < L && c < nchunk) out[(long)i * nchunk + c] = tile[threadIdx.x][r];
  }
}

// [i][j][chunk] (one problem of a materialising batch run) -> the reference's
// [n][j] storage with its index conventions (u one column earlier, cholesky.h:131-151).
__global__ void __launch_bounds__(256) deinterleave_factor_kernel(
    const double* __restrict__ phi_i, const double* __restrict__ u_i,
    const double* __restrict__ W_i, const double* __restrict__ D_i, double* __restrict__ phi,
    double* __restrict__ u, double* __restrict__ W, double* __restrict__ D, int N, int J, int L,
    int nchunk) {
  const long total = (long)N * J;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / J), j = (int)(e % J);
    const int c = n / L, i = n % L;
    const long src = ((long)i * J + j) * nchunk + c;
    W[e] = W_i[src];
    if (n + 1 < N) phi[e] = phi_i[src];
    if (n >= 1) u[e - J] = u_i[src];
    if (j == 0) D[n] = D_i[(long)i * nchunk + c];
  }
}

void launch_deinterleave_factor(const double* phi_i, const double* u_i, const double* W_i,
                                const double* D_i, double* phi, double* u, double* W, double* D,
                                int N, int J, int L, int nchunk, hipStream_t s) {
  const long total = (long)N * J;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(deinterleave_factor_kernel, dim3(blocks), dim3(256), 0, s, phi_i, u_i, W_i,
                     D_i, phi, u, W, D, N, J, L, nchunk);
}

// The warm kernel's copy: [row][chunk] per problem with row r of chunk c = sample n = c L - Kpad + r, rows =
// Kpad + L + 8 (every chunk's column starts with the Kpad samples in front of it -- its warm-up -- and ends with the
// look-ahead of its last steps).  Samples outside the series are padding a recurrence step ignores: before the
// start t = t_0, after the end t = t_{N-1} (a zero time step), the diagonal 1e300 (1 / D ~ 0), y = 0.
__global__ void __launch_bounds__(256) relayout_warm_kernel(const double* __restrict__ src, long src_stride,
                                                            double* __restrict__ dst, long dst_stride, int N, int L,
                                                            int nchunk, int Kpad, int rows, int pad_kind) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const double* in = src + (long)b * src_stride;
  double* out = dst + (long)b * dst_stride;
  const double pad_lo = pad_kind == 1 ? in[0] : (pad_kind == 2 ? 1e300 : 0.0);
  const double pad_hi = pad_kind == 1 ? in[N - 1] : (pad_kind == 2 ? 1e300 : 0.0);
  for (int q = threadIdx.y; q < 32; q += 8) {
    const int c = c0 + q, r = r0 + threadIdx.x;
    const long n = (long)c * L - Kpad + r;
    double v = n < 0 ? pad_lo : pad_hi;
    if (c < nchunk && r < rows && n >= 0 && n < N) v = in[n];
    tile[q][threadIdx.x] = v;
  }
  __syncthreads();
  for (int q = threadIdx.y; q < 32; q += 8) {
    const int r = r0 + q, c = c0 + threadIdx.x;
    if (r < rows && c < nchunk) out[(long)r * nchunk + c] = tile[threadIdx.x][q];
  }
}

void launch_relayout_warm(const double* src, long src_stride, double* dst, long dst_stride, int nsrc, int N, int L,
                          int nchunk, int Kpad, int rows, int pad_kind, hipStream_t s) {
  dim3 grid((rows + 31) / 32, (nchunk + 31) / 32, nsrc);
  hipLaunchKernelGGL(relayout_warm_kernel, grid, dim3(32, 8), 0, s, src, src_stride, dst, dst_stride, N, L, nchunk,
                     Kpad, rows, pad_kind);
}

void launch_relayout(const double* src, long src_stride, double* dst, long dst_stride, int nsrc,
                     int N, int L, int nchunk, int pad_kind, hipStream_t s) {
  dim3 grid((L + 31) / 32, (nchunk + 31) / 32, nsrc);
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(32, 8), 0, s, src, src_stride, dst, dst_stride,
                     N, L, nchunk, pad_kind);
}
// the inverse of relayout_kernel: [problem][i][chunk] -> [problem][n] for n = chunk * L + i < N (clr_batch_solve's result)
__global__ void __launch_bounds__(256) relayout_back_kernel(const double* __restrict__ src, long src_stride,
                                                            double* __restrict__ dst, long dst_stride, int N, int L, int nchunk) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z, i0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const double* in = src + (long)b * src_stride;
  double* out = dst + (long)b * dst_stride;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = i0 + r, c = c0 + threadIdx.x;
    tile[r][threadIdx.x] = (i < L && c < nchunk) ? in[(long)i * nchunk + c] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int c = c0 + r, i = i0 + threadIdx.x;
    const long n = (long)c * L + i;
    if (c < nchunk && i < L && n < N) out[n] = tile[threadIdx.x][r];
  }
}
void launch_relayout_back(const double* src, long src_stride, double* dst, long dst_stride, int nsrc, int N, int L, int nchunk,
                          hipStream_t s) {
  dim3 grid((L + 31) / 32, (nchunk + 31) / 32, nsrc);
  hipLaunchKernelGGL(relayout_back_kernel, grid, dim3(32, 8), 0, s, src, src_stride, dst, dst_stride, N, L, nchunk);
}

// ---- re-planning of a few problems as a small plan of their own (api_batch.hip: rescue_run) ---------------------
// series of the problems idx[0..n) of a plan -> rows 0..n of another plan's arrays (device to device)
__global__ void __launch_bounds__(256) gather_series_kernel(const double* __restrict__ src, long src_stride,
                                                            double* __restrict__ dst, const int* __restrict__ idx, int N) {
  const int k = blockIdx.y;
  const double* in = src + (long)idx[k] * src_stride;
  double* out = dst + (long)k * N;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) out[i] = in[i];
}
void launch_gather_series(const double* src, long src_stride, double* dst, const int* idx, int n, int N, hipStream_t s) {
  hipLaunchKernelGGL(gather_series_kernel, dim3(std::min((N + 255) / 256, 64), n), dim3(256), 0, s, src, src_stride, dst, idx, N);
}
// coefficient tables a_real c_real [B][JR] | a_comp b_comp c_comp d_comp [B][JC] | jitter [B] -> the same layout for n problems
__global__ void __launch_bounds__(64) gather_coeffs_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                           const int* __restrict__ idx, int B, int n, int JR, int JC) {
  const int k = blockIdx.x, b = idx[k], per = 2 * JR + 4 * JC + 1;
  for (int e = threadIdx.x; e < per; e += 64) {
    long so, dofs;
    if (e < 2 * JR) { const int blk = e / JR, j = e % JR; so = (long)blk * B * JR + (long)b * JR + j; dofs = (long)blk * n * JR + (long)k * JR + j; }
    else if (e < 2 * JR + 4 * JC) {
      const int q = e - 2 * JR, blk = q / JC, j = q % JC;
      so = 2L * B * JR + (long)blk * B * JC + (long)b * JC + j;
      dofs = 2L * n * JR + (long)blk * n * JC + (long)k * JC + j;
    } else { so = 2L * B * JR + 4L * B * JC + b; dofs = 2L * n * JR + 4L * n * JC + k; }
    dst[dofs] = src[so];
  }
}
void launch_gather_coeffs(const double* src, double* dst, const int* idx, int B, int n, int JR, int JC, hipStream_t s) {
  hipLaunchKernelGGL(gather_coeffs_kernel, dim3(n), dim3(64), 0, s, src, dst, idx, B, n, JR, JC);
}
// results (ll | logdet | quad | status) and route of the n re-planned problems -> their places in the parent plan
__global__ void __launch_bounds__(64) scatter_results_kernel(const double* __restrict__ sub_out, const int* __restrict__ sub_level,
                                                             int n, double* __restrict__ out, int* __restrict__ level,
                                                             int B, const int* __restrict__ idx) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const int b = idx[k];
  out[b] = sub_out[k];
  out[(long)B + b] = sub_out[(long)n + k];
  out[2L * B + b] = sub_out[2L * n + k];
  reinterpret_cast<int*>(out + 3L * B)[b] = reinterpret_cast<const int*>(sub_out + 3L * n)[k];
  const int lv = sub_level[k];
  level[b] = lv < 1 ? 1 : lv;  // (checked chunked replay at least; 2: the sub-plan's sequential recurrence settled it)
}
void launch_scatter_results(const double* sub_out, const int* sub_level, int n, double* out, int* level, int B, const int* idx,
                            hipStream_t s) {
  hipLaunchKernelGGL(scatter_results_kernel, dim3((n + 63) / 64), dim3(64), 0, s, sub_out, sub_level, n, out, level, B, idx);
}

// Which compute units a stream's workgroups land on (diagnostic for CU-masked streams): every workgroup marks the
// (XCC, HW_ID cu / sh / se) it ran on; `spin` iterations of dependent arithmetic keep it resident long enough for the
// dispatcher to spread a grid over every unit the stream may use.
__global__ void __launch_bounds__(64) cu_census_kernel(int* seen, int spin) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7;    // HW_REG_XCC_ID
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  double x = 1.0 + threadIdx.x * 1e-9;
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
  if (threadIdx.x == 0) atomicAdd(seen + xcc * 256 + ((hw >> 8) & 0xff), x > 0.0 ? 1 : 0);
}
void launch_cu_census(int* seen, int blocks, int spin, hipStream_t s) {
  hipLaunchKernelGGL(cu_census_kernel, dim3(blocks), dim3(64), 0, s, seen, spin);
}

}  // namespace clr
