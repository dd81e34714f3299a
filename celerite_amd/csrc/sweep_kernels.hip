// celerite_amd/csrc/sweep_kernels.hip -- dot_solve / solve on a stored factor as chunked
// scans over n (single-solver API, long series).
//
// The reference's sweeps (cholesky.h:236-260, :343-357) are one dependent chain of N
// steps:   f <- p o (f + g x_prev) ;  x = in - h . f
// with (p, g, h, in) = (phi, W, u, b) going forward and (phi, u, W, x / D) going
// backward.  On the augmented state z = (f, x) in R^(J+1) a step is AFFINE,
//     f' = p o f + (p o g) x ,   x' = in - h . f' ,
// so a chunk of L steps composes into z_end = A z_start + c (A is (J+1) x (J+1)) and
// the sweep becomes the usual three phases:
//   summarize  one lane per chunk folds its steps into (A, c)      -- parallel
//   prefix     z at every chunk start, (J+1)^2 flops per chunk     -- sequential, tiny
//   replay     the reference recurrence itself from the known start -- parallel
// ("dot_solve / solve are affine-linear once the factor exists").
// A lane reads its own run of the factor: J contiguous doubles per array and step (one
// 64-B line at J = 8).  Widths 1..8 (compile-time); wider factors and short series keep
// the sequential kernels of generic_kernels.hip (N < 256 or J > 8).  Results differ from the sequential
// sweep only by re-association (tests: <= 1e-12 relative against the oracle).
#include "clr_generic_kernels.h"

namespace clr {

namespace {

template <int J>
struct StepData {
  double p[J], g[J], h[J];
  double in, d;
};

// step s = 1 .. N-1 of the sweep (forward: sample s; backward: sample N-1-s)
template <int J>
__device__ __forceinline__ int load_step(const SweepParams& P, const double* in, int s, StepData<J>& st) {
  const int n = P.backward ? P.N - 1 - s : s;
  const long col = (long)J * (P.backward ? n : s - 1);
  const double* gp = P.backward ? P.u : P.W;
  const double* hp = P.backward ? P.W : P.u;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    st.p[j] = P.phi[col + j];
    st.g[j] = gp[col + j];
    st.h[j] = hp[col + j];
  }
  st.d = P.D[n];
  st.in = P.backward ? in[n] / st.d : in[n];  // cholesky.h:249 folded into the backward input
  return n;
}

template <int J>
__global__ void __launch_bounds__(64) sweep_summarize_kernel(const SweepParams P) {
  constexpr int K = J + 1;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const double* in = P.in + (long)blockIdx.y * P.N;
  double A[K][K], cv[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
#pragma unroll
    for (int j = 0; j < K; ++j) A[i][j] = (i == j) ? 1.0 : 0.0;
    cv[i] = 0.0;
  }
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  for (int s = s0; s < s1; ++s) {
    StepData<J> st;
    load_step<J>(P, in, s, st);
    // rows f_i: p_i row_i + (p_i g_i) row_x ; row x: -sum_i h_i row_i' (+ in for c)
#pragma unroll
    for (int col = 0; col < K; ++col) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < J; ++i) {
        A[i][col] = st.p[i] * (A[i][col] + st.g[i] * A[J][col]);
        acc += st.h[i] * A[i][col];
      }
      A[J][col] = -acc;
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) {
      cv[i] = st.p[i] * (cv[i] + st.g[i] * cv[J]);
      acc += st.h[i] * cv[i];
    }
    cv[J] = st.in - acc;
  }
  double* o = P.elems + ((long)blockIdx.y * P.nchunk + c) * (K * K + K);
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) o[i * K + j] = A[i][j];
#pragma unroll
  for (int i = 0; i < K; ++i) o[K * K + i] = cv[i];
}

// one thread per right-hand side: z at the start of every chunk
template <int J>
__global__ void __launch_bounds__(64) sweep_prefix_kernel(const SweepParams P) {
  constexpr int K = J + 1;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= P.nrhs) return;
  const double* in = P.in + (long)col * P.N;
  double z[K];
#pragma unroll
  for (int i = 0; i < J; ++i) z[i] = 0.0;
  // the sweep's first sample: x_0 = b_0 (cholesky.h:238) / x_{N-1} / D_{N-1} (:249,251)
  z[J] = P.backward ? in[P.N - 1] / P.D[P.N - 1] : in[0];
  for (int c = 0; c < P.nchunk; ++c) {
    double* st = P.starts + ((long)col * P.nchunk + c) * K;
#pragma unroll
    for (int i = 0; i < K; ++i) st[i] = z[i];
    const double* e = P.elems + ((long)col * P.nchunk + c) * (K * K + K);
    double nz[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      double acc = e[K * K + i];
#pragma unroll
      for (int j = 0; j < K; ++j) acc += e[i * K + j] * z[j];
      nz[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) z[i] = nz[i];
  }
}

template <int J>
__global__ void __launch_bounds__(64) sweep_replay_kernel(const SweepParams P) {
  constexpr int K = J + 1;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const double* in = P.in + (long)blockIdx.y * P.N;
  double* out = P.out ? P.out + (long)blockIdx.y * P.N : nullptr;
  const double* st0 = P.starts + ((long)blockIdx.y * P.nchunk + c) * K;
  double f[J], x = st0[J], quad = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) f[i] = st0[i];
  if (c == 0) {  // the first sample of the sweep belongs to chunk 0
    const int n0 = P.backward ? P.N - 1 : 0;
    if (out) out[n0] = x;
    quad = x * (x / P.D[n0]);  // cholesky.h:347
  }
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  for (int s = s0; s < s1; ++s) {
    StepData<J> st;
    const int n = load_step<J>(P, in, s, st);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) {  // cholesky.h:243-246 / :255-258 / :350-354
      f[i] = st.p[i] * (f[i] + st.g[i] * x);
      acc += st.h[i] * f[i];
    }
    x = st.in - acc;
    if (out) out[n] = x;
    quad += x * x / st.d;  // :356
  }
  if (P.part) P.part[(long)blockIdx.y * P.nchunk + c] = quad;
}

// dot_solve: chunk partials summed in chunk order, one thread per right-hand side
__global__ void __launch_bounds__(64) sweep_finalize_kernel(const SweepParams P) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= P.nrhs) return;
  double q = 0.0;
  for (int c = 0; c < P.nchunk; ++c) q += P.part[(long)col * P.nchunk + c];
  P.quad[col] = q;
}

template <int J>
void run(const SweepParams& P, hipStream_t s) {
  const dim3 grid((P.nchunk + 63) / 64, P.nrhs);
  hipLaunchKernelGGL((sweep_summarize_kernel<J>), grid, dim3(64), 0, s, P);
  hipLaunchKernelGGL((sweep_prefix_kernel<J>), dim3((P.nrhs + 63) / 64), dim3(64), 0, s, P);
  hipLaunchKernelGGL((sweep_replay_kernel<J>), grid, dim3(64), 0, s, P);
  if (P.quad) hipLaunchKernelGGL(sweep_finalize_kernel, dim3((P.nrhs + 63) / 64), dim3(64), 0, s, P);
}

// ---------------------------------------------------------------------------
// dot_L (cholesky.h:409-431): no feedback -- f_n = phi o (f_{n-1} + W t_{n-1}) with
// t_n = sqrt(D_n) z_n, y_n = t_n + u . f_n: a DIAGONAL affine recurrence, element (a, c).
// ---------------------------------------------------------------------------
template <int J, bool REPLAY>
__global__ void __launch_bounds__(64) dotl_kernel(const SweepParams P) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= P.nchunk) return;
  const double* z = P.in + (long)blockIdx.y * P.N;
  double* y = P.out + (long)blockIdx.y * P.N;
  double* e = P.elems + ((long)blockIdx.y * P.nchunk + c) * (2 * J);
  double a[J], f[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { a[j] = 1.0; f[j] = REPLAY ? P.starts[((long)blockIdx.y * P.nchunk + c) * J + j] : 0.0; }
  const int s0 = c * P.L + 1;
  const int s1 = min(s0 + P.L, P.N);
  if (REPLAY && c == 0) y[0] = sqrt(P.D[0]) * z[0];  // :421-422
  for (int n = s0; n < s1; ++n) {
    const long col = (long)J * (n - 1);
    const double tprev = sqrt(P.D[n - 1]) * z[n - 1];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {  // :424-426
      const double ph = P.phi[col + j];
      f[j] = ph * (f[j] + P.W[col + j] * tprev);
      if (!REPLAY) a[j] *= ph;
      acc += P.u[col + j] * f[j];
    }
    if (REPLAY) y[n] = sqrt(P.D[n]) * z[n] + acc;
  }
  if (!REPLAY) {
#pragma unroll
    for (int j = 0; j < J; ++j) { e[j] = a[j]; e[J + j] = f[j]; }
  }
}

template <int J>
__global__ void __launch_bounds__(64) dotl_prefix_kernel(const SweepParams P) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= P.nrhs) return;
  double f[J];
#pragma unroll
  for (int j = 0; j < J; ++j) f[j] = 0.0;
  for (int c = 0; c < P.nchunk; ++c) {
    const double* e = P.elems + ((long)col * P.nchunk + c) * (2 * J);
    double* st = P.starts + ((long)col * P.nchunk + c) * J;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      st[j] = f[j];
      f[j] = e[j] * f[j] + e[J + j];
    }
  }
}

template <int J>
void run_dotl(const SweepParams& P, hipStream_t s) {
  const dim3 grid((P.nchunk + 63) / 64, P.nrhs);
  hipLaunchKernelGGL((dotl_kernel<J, false>), grid, dim3(64), 0, s, P);
  hipLaunchKernelGGL((dotl_prefix_kernel<J>), dim3((P.nrhs + 63) / 64), dim3(64), 0, s, P);
  hipLaunchKernelGGL((dotl_kernel<J, true>), grid, dim3(64), 0, s, P);
}

// ---------------------------------------------------------------------------
// predict (cholesky.h:599-698): two diagonal recurrences over the samples,
//   forward   Q_r <- (Q_r + alpha_n v_r(t_n)) exp(-c_r (t_{n+1} - t_n))       (:618-636)
//   backward  Q_r <- (Q_r + alpha_n u_r(t_n)) exp(-c_r (t_n - t_{n-1}))       (:662-679)
// and every prediction point x_m reads the forward Q of the sample interval it falls in
// (t_n < x_m <= t_{n+1}; n = N-1 beyond the data; none at or before t_0, :616,638) and
// the backward Q of t_{n-1} < x_m <= t_n (n = 0 at or before t_0; none beyond the
// data, :657,681).  Chunk starts of both recurrences come from a chunked scan; then ONE
// THREAD PER PREDICTION POINT walks from its chunk start to its interval (<= L steps)
// and evaluates both sums.  Needs the prediction points sorted (the reference's while
// loops assume it); api_solver.hip checks and otherwise uses the sequential kernel.
// ---------------------------------------------------------------------------
constexpr int PJ = 8;  // rows (J_real + 2 J_comp) handled per thread

struct PredictRows {
  double a[PJ], b[PJ], c[PJ], d[PJ];
  int kind[PJ];  // 0 real, 1 cos row, 2 sin row, -1 unused
  int rows;
};

// rows row0 .. row0 + PJ - 1 of the J_real + 2 J_comp rows (wider kernels go block of PJ rows by block: the
// recurrences are diagonal, every row is on its own)
__device__ __forceinline__ void load_rows(const GenericProblem& g, PredictRows& r, int row0 = 0) {
  r.rows = g.J_real + 2 * g.J_comp;
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int row = row0 + j;
    r.kind[j] = -1; r.a[j] = r.b[j] = r.c[j] = r.d[j] = 0.0;
    if (row < g.J_real) {
      r.kind[j] = 0; r.a[j] = g.a_real[row]; r.c[j] = g.c_real[row];
    } else if (row < r.rows) {
      const int jj = (row - g.J_real) >> 1;
      r.kind[j] = 1 + ((row - g.J_real) & 1);
      r.a[j] = g.a_comp[jj]; r.b[j] = g.b_comp[jj]; r.c[j] = g.c_comp[jj]; r.d[j] = g.d_comp[jj];
    }
  }
}
// v_r(t): 1 | cos | sin (:622,627,629);  u_r(t): a | a cos + b sin | a sin - b cos (:666,672-677)
__device__ __forceinline__ void row_uv(const PredictRows& r, int j, double t, double* u, double* v) {
  if (r.kind[j] <= 0) { *u = r.a[j]; *v = 1.0; return; }
  double sd, cd;
  sincos(r.d[j] * t, &sd, &cd);
  *v = r.kind[j] == 1 ? cd : sd;
  *u = r.kind[j] == 1 ? r.a[j] * cd + r.b[j] * sd : r.a[j] * sd - r.b[j] * cd;
}
// one step of the forward (dir = 0) or backward (dir = 1) recurrence at sample n
__device__ __forceinline__ void predict_step(const PredictRows& r, const double* t, const double* alpha,
                                             int N, int n, int dir, double* Q, double* A) {
  const double tn = t[n];
  const double dt = dir == 0 ? ((n < N - 1) ? t[n + 1] - tn : 0.0) : ((n > 0) ? tn - t[n - 1] : 0.0);
  const double al = alpha[n];
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    if (r.kind[j] < 0) continue;
    double u, v;
    row_uv(r, j, tn, &u, &v);
    const double e = exp(-r.c[j] * dt);
    Q[j] = (Q[j] + al * (dir == 0 ? v : u)) * e;
    if (A) A[j] *= e;
  }
}

// elems: [row block][dir][chunk][2 PJ] = (a, c);  starts: [row block][dir][chunk][PJ]
__global__ void __launch_bounds__(64) predict_summarize_kernel(GenericProblem g, const double* alpha,
                                                               int nchunk, int L, double* elems) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchunk) return;
  const int dir = blockIdx.y, blk = blockIdx.z;
  elems += (long)blk * 2 * nchunk * (2 * PJ);
  PredictRows r;
  load_rows(g, r, blk * PJ);
  double Q[PJ], A[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) { Q[j] = 0.0; A[j] = 1.0; }
  const int lo = c * L, hi = min(lo + L, g.N);
  if (dir == 0) for (int n = lo; n < hi; ++n) predict_step(r, g.t, alpha, g.N, n, 0, Q, A);
  else          for (int n = hi - 1; n >= lo; --n) predict_step(r, g.t, alpha, g.N, n, 1, Q, A);
  double* e = elems + ((long)dir * nchunk + c) * (2 * PJ);
#pragma unroll
  for (int j = 0; j < PJ; ++j) { e[j] = A[j]; e[PJ + j] = Q[j]; }
}

// The chunk start states of predict's two diagonal recurrences: Q' = a Q + q per row, an affine map per chunk -- composed
// per thread over a contiguous run of chunks, scanned over the 256 threads (Kogge-Stone on (a, q) pairs through LDS), then
// every thread walks its run again from the state the scan found for it.  Workgroup = (row block, direction).
// (Round 4; rounds 2-3 walked the chunks with ONE thread per direction: 0.30 ms of predict's 2.1 ms at N = 1e5.)
__global__ void __launch_bounds__(256) predict_prefix_kernel(int nchunk, const double* elems, double* starts) {
  __shared__ double sa[PJ][256], sq[PJ][256];
  const int tid = threadIdx.x, dir = blockIdx.y, blk = blockIdx.x;
  elems += ((long)blk * 2 + dir) * nchunk * (2 * PJ);
  starts += ((long)blk * 2 + dir) * nchunk * PJ;
  const int S = (nchunk + 255) / 256, p0 = tid * S, p1 = (p0 + S < nchunk) ? p0 + S : nchunk;
  auto chunk_at = [&](int p) { return dir == 0 ? p : nchunk - 1 - p; };  // the backward recurrence meets the chunks in reverse
  double A[PJ], Q[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) { A[j] = 1.0; Q[j] = 0.0; }
  for (int p = p0; p < p1; ++p) {
    const double* e = elems + (long)chunk_at(p) * (2 * PJ);
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      Q[j] = fma(e[j], Q[j], e[PJ + j]);
      A[j] *= e[j];
    }
  }
#pragma unroll
  for (int j = 0; j < PJ; ++j) { sa[j][tid] = A[j]; sq[j][tid] = Q[j]; }
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // inclusive scan: (a1, q1) then (a2, q2) = (a2 a1, a2 q1 + q2)
    double a1[PJ], q1[PJ];
    if (tid >= d) {
#pragma unroll
      for (int j = 0; j < PJ; ++j) { a1[j] = sa[j][tid - d]; q1[j] = sq[j][tid - d]; }
    }
    __syncthreads();
    if (tid >= d) {
#pragma unroll
      for (int j = 0; j < PJ; ++j) {
        sq[j][tid] = fma(sa[j][tid], q1[j], sq[j][tid]);
        sa[j][tid] *= a1[j];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < PJ; ++j) Q[j] = tid > 0 ? sq[j][tid - 1] : 0.0;  // the state before this thread's run
  for (int p = p0; p < p1; ++p) {
    const int c = chunk_at(p);
    const double* e = elems + (long)c * (2 * PJ);
    double* st = starts + (long)c * PJ;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      st[j] = Q[j];
      Q[j] = fma(e[j], Q[j], e[PJ + j]);
    }
  }
}

__global__ void __launch_bounds__(64) predict_points_kernel(GenericProblem g, const double* alpha, int nchunk,
                                                            int L, const double* starts, int M,
                                                            const double* xs, double* pred) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= M) return;
  const int N = g.N;
  const double* t = g.t;
  const double xm = xs[m];
  // k = number of samples with t_n < x_m  (lower bound)
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (t[mid] < xm) lo = mid + 1; else hi = mid;
  }
  const int k = lo;
  double total = 0.0;
  const int nblk = (g.J_real + 2 * g.J_comp + PJ - 1) / PJ;
  for (int blk = 0; blk < nblk; ++blk) {
    PredictRows r;
    load_rows(g, r, blk * PJ);
    const double* bstarts = starts + (long)blk * 2 * nchunk * PJ;
    if (k >= 1) {  // forward: interval n = k - 1 (t_n < x_m <= t_{n+1}, or beyond the last sample)
      const int nf = k - 1, c = nf / L;
      double Q[PJ];
#pragma unroll
      for (int j = 0; j < PJ; ++j) Q[j] = bstarts[((long)0 * nchunk + c) * PJ + j];
      for (int n = c * L; n <= nf; ++n) predict_step(r, t, alpha, N, n, 0, Q, nullptr);
      const double tref = nf < N - 1 ? t[nf + 1] : t[N - 1];
      const double dt = xm - tref;  // :640
#pragma unroll
      for (int j = 0; j < PJ; ++j) {
        if (r.kind[j] < 0) continue;
        double u, v;
        row_uv(r, j, xm, &u, &v);
        total += u * exp(-r.c[j] * dt) * Q[j];  // :643,649-650
      }
    }
    if (k < N) {  // backward: interval n = k (t_{n-1} < x_m <= t_n, or at / before the first sample)
      const int nb = k, c = nb / L;
      double Q[PJ];
#pragma unroll
      for (int j = 0; j < PJ; ++j) Q[j] = bstarts[((long)1 * nchunk + c) * PJ + j];
      for (int n = min(c * L + L, N) - 1; n >= nb; --n) predict_step(r, t, alpha, N, n, 1, Q, nullptr);
      const double tref = nb > 0 ? t[nb - 1] : t[0];
      const double dt = tref - xm;  // :683
      double pm = 0.0;
#pragma unroll
      for (int j = 0; j < PJ; ++j) {
        if (r.kind[j] < 0) continue;
        double u, v;
        row_uv(r, j, xm, &u, &v);
        pm += v * exp(-r.c[j] * dt) * Q[j];  // :686,690-691
      }
      total += pm;
    }
  }
  pred[m] = total;
}

}  // namespace

// (any width: rows go in blocks of PJ; the sequential kernel took 168 ms at width 16, N = 1e5, M = 2e4 -- the CPU 40)
bool predict_scan_supported(int N, int J_real, int J_comp) { return J_real + 2 * J_comp >= 1 && N >= 256; }
static int predict_blocks(const GenericProblem& g) { return (g.J_real + 2 * g.J_comp + PJ - 1) / PJ; }
size_t predict_workspace_doubles(int nchunk, int rows) { return (size_t)((rows + PJ - 1) / PJ) * 2 * nchunk * (3 * PJ); }

void launch_predict_scan(const GenericProblem& g, const double* alpha, int M, const double* xs, double* pred,
                         double* workspace, int nchunk, int L, hipStream_t s) {
  const int nblk = predict_blocks(g);
  double* elems = workspace;
  double* starts = workspace + (size_t)nblk * 2 * nchunk * 2 * PJ;
  hipLaunchKernelGGL(predict_summarize_kernel, dim3((nchunk + 63) / 64, 2, nblk), dim3(64), 0, s, g, alpha, nchunk, L, elems);
  hipLaunchKernelGGL(predict_prefix_kernel, dim3(nblk, 2), dim3(256), 0, s, nchunk, elems, starts);
  hipLaunchKernelGGL(predict_points_kernel, dim3((M + 63) / 64), dim3(64), 0, s, g, alpha, nchunk, L, starts, M, xs, pred);
}

void launch_dot_L_scan(SweepParams P, double* workspace, hipStream_t s) {
  const size_t pc = (size_t)P.nrhs * P.nchunk;
  P.elems = workspace;
  P.starts = P.elems + pc * 2 * P.J;
  switch (P.J) {
    case 1: run_dotl<1>(P, s); break;
    case 2: run_dotl<2>(P, s); break;
    case 3: run_dotl<3>(P, s); break;
    case 4: run_dotl<4>(P, s); break;
    case 5: run_dotl<5>(P, s); break;
    case 6: run_dotl<6>(P, s); break;
    case 7: run_dotl<7>(P, s); break;
    case 8: run_dotl<8>(P, s); break;
    default: break;
  }
}

// (a sequential sweep costs ~0.37 us per sample; three launches of the scan ~15 us)
bool sweep_scan_supported(int N, int J) { return J >= 1 && J <= 8 && N >= 256; }

// ~1.8 sqrt(N) chunks balance the parallel phases (0.65 us per step) against the
// sequential prefix (0.2 us per chunk); a multiple of 64 lanes
int sweep_chunks(int N) {
  long nc = (long)(1.8 * sqrt((double)N));
  nc = ((nc + 63) / 64) * 64;
  const long maxc = std::max<long>(1, (N - 1) / 8);
  if (nc > maxc) nc = maxc;
  return (int)std::max<long>(nc, 1);
}

size_t sweep_workspace_doubles(int J, int nchunk, int nrhs) {
  const size_t K = (size_t)J + 1;
  return (size_t)nrhs * nchunk * (K * K + K + K + 1);
}

void launch_sweep_scan(SweepParams P, double* workspace, hipStream_t s) {
  const size_t K = (size_t)P.J + 1, pc = (size_t)P.nrhs * P.nchunk;
  P.elems = workspace;
  P.starts = P.elems + pc * (K * K + K);
  P.part = P.starts + pc * K;
  switch (P.J) {
    case 1: run<1>(P, s); break;
    case 2: run<2>(P, s); break;
    case 3: run<3>(P, s); break;
    case 4: run<4>(P, s); break;
    case 5: run<5>(P, s); break;
    case 6: run<6>(P, s); break;
    case 7: run<7>(P, s); break;
    case 8: run<8>(P, s); break;
    default: break;
  }
}

}  // namespace clr
