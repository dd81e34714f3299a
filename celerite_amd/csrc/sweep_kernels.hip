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
// the sequential kernels of generic_kernels.hip.  Results differ from the sequential
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

}  // namespace

bool sweep_scan_supported(int N, int J) { return J >= 1 && J <= 8 && N >= 2048; }

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
