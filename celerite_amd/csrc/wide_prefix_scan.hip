// celerite_amd/csrc/wide_prefix_scan.hip -- the prefix of the wide scan (padded widths 16 / 32) as a PARALLEL scan.
//
// wide_prefix32_kernel / prefix_coop_kernel<16, 16> (wide_kernels.hip) walk a problem's chunks one after the other:
// ~50 us resp. ~14 us per chunk on one workgroup while the rest of the chip idles -- for ONE long series (the reference's
// object API, cholesky.h:41-210 called once per light curve) that walk is the largest part of the call.  The chunk elements
// compose in closed form (clr_core.h: compose_elements -- the formulas below are that function's, at the padded width,
// on a workgroup), so few problems with many chunks get a Kogge-Stone scan instead: ceil(log2(nchunk - 1)) launches,
// launch k composing every element c >= 2^k with the element 2^k places before it, all (problem, c) side by side.
//
//   e12 = "e1, then e2":   Mi = (I + C1 Jm2)^-1 ,  w = eta2 - Jm2 b1
//     C12 = C2 + A2 (Mi C1) A2^T      b12   = b2 + A2 Mi (b1 + C1 eta2)
//     A12 = A2 (Mi A1)                eta12 = eta1 + (Mi A1)^T w          Jm12 = Jm1 + sym(A1^T Jm2 (Mi A1))
//
// A window that reaches back to chunk 0 is FINAL: its (C, b) is the start state of the next chunk (the zero state pushed
// through the window) and nothing will ever be composed in front of it, so its riders (A, eta, Jm) are not formed and it
// lives in P.starts; later levels read it from there as their e1.  Chunk 0 itself may come from the riderless summarize
// (BatchParams::L0): only its (C, b) is read.  Elimination: Gauss-Jordan with partial pivoting on
// [ I + C1 Jm2 | C1 | A1 | b1 + C1 eta2 ] in LDS, 256 threads -- wide_correct_kernel's, with J more right-hand sides.
#include "clr_batch_kernels.h"
#include "clr_options.h"

#include <stdlib.h>

namespace clr {
namespace {

struct ScanLevel {
  const double* in;   // level input: [B][nchunk][ELEM] (level 0: the summarize's own elements)
  double* out;        // level output (non-final windows only)
  double* starts;     // [B][nchunk][START]
  int* need_exact;    // [B]: cleared by the first level (as the sequential prefix kernels do)
  int B, nchunk, d;
};

template <int J>
__global__ void __launch_bounds__(256) wide_scan_level_kernel(const ScanLevel K) {
  constexpr int SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  constexpr int LD = J + 1, NCMAX = 3 * J + 1, LT = NCMAX + 1, NT = 256;
  __shared__ double A1m[J * LD], A2m[J * LD], J2m[J * LD], Ym[J * LD], T[J * LT];
  __shared__ double b1v[J], b2v[J], e1v[J], e2v[J], wv[J];
  const int tid = threadIdx.x, lane = tid & 63;
  const int per = K.nchunk - 1 - K.d + (K.d == 1 ? 1 : 0);  // blocks per problem; level 0 has one prologue block more
  const int prob = blockIdx.x / per, idx = blockIdx.x % per;
  if (K.d == 1 && idx == per - 1) {
    // prologue: chunk 0's (C, b) is the start state of chunk 1
    const double* E0 = K.in + (long)prob * K.nchunk * ELEM;
    double* o = K.starts + ((long)prob * K.nchunk + 1) * START;
    for (int i = tid; i < SZ; i += NT) o[i] = E0[J * J + J + i];
    if (tid < J) o[SZ + tid] = E0[J * J + tid];
    if (tid == 0) K.need_exact[prob] = 0;
    return;
  }
  const int c = K.d + idx;                 // d <= c < nchunk - 1
  const bool final_ = c - K.d < K.d;       // e1 reaches back to chunk 0
  const double* E2 = K.in + ((long)prob * K.nchunk + c) * ELEM;
  const double *A2 = E2, *b2 = E2 + J * J, *C2 = b2 + J, *eta2 = C2 + SZ, *Jm2 = eta2 + J;
  const double* E1 = K.in + ((long)prob * K.nchunk + (c - K.d)) * ELEM;
  const double *A1 = E1, *eta1 = E1 + J * J + J + SZ, *Jm1 = eta1 + J;
  const double *C1, *b1;
  if (final_ && K.d > 1) {                 // a final window lives in the start states
    C1 = K.starts + ((long)prob * K.nchunk + (c - K.d) + 1) * START;
    b1 = C1 + SZ;
  } else {
    C1 = E1 + J * J + J;
    b1 = E1 + J * J;
  }
  const int NC = final_ ? 2 * J + 1 : 3 * J + 1;
  const int CH = NC - 1;                   // column of h = b1 + C1 eta2

  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    A2m[r * LD + q] = A2[i];
    J2m[r * LD + q] = Jm2[sym(r, q)];
    T[r * LT + J + q] = C1[sym(r, q)];
    if (!final_) { A1m[r * LD + q] = A1[i]; T[r * LT + 2 * J + q] = A1[i]; }
  }
  if (tid < J) { b1v[tid] = b1[tid]; b2v[tid] = b2[tid]; e2v[tid] = eta2[tid]; e1v[tid] = final_ ? 0.0 : eta1[tid]; }
  __syncthreads();
  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    double acc = (r == q) ? 1.0 : 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(T[r * LT + J + k], J2m[k * LD + q], acc);
    T[r * LT + q] = acc;
  }
  if (tid < J) {
    double h = b1v[tid];
#pragma unroll 8
    for (int k = 0; k < J; ++k) h = fma(T[tid * LT + J + k], e2v[k], h);
    T[tid * LT + CH] = h;
  } else if (tid >= 64 && tid < 64 + J) {  // w = eta2 - Jm2 b1
    const int r = tid - 64;
    double acc = e2v[r];
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(-J2m[r * LD + k], b1v[k], acc);
    wv[r] = acc;
  }
  __syncthreads();

  // Gauss-Jordan with partial pivoting: wave w owns rows RW w .. RW w + RW - 1 of the tableau, lane = column (stride 64)
  constexpr int RW = J / 4;
  const int wave = tid >> 6;
  for (int col = 0; col < J; ++col) {
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
    __syncthreads();  // (every wave has finished its search of column `col`)
    // row swap + scaling of the pivot row: thread = column (the pivot itself, column `col`, stays unscaled: it is never
    // read again)
    const double inv = 1.0 / T[piv * LT + col];
    __syncthreads();  // (everybody has read the pivot)
    if (tid < NC) {
      const double top = T[piv * LT + tid], old = T[col * LT + tid];
      if (piv != col) T[piv * LT + tid] = old;
      T[col * LT + tid] = tid > col ? top * inv : top;
    }
    __syncthreads();
    double m[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) m[r] = T[(RW * wave + r) * LT + col];
    for (int cc = lane; cc < NC; cc += 64) {
      if (cc > col) {
        const double t = T[col * LT + cc];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int row = RW * wave + r;
          if (row != col) T[row * LT + cc] = fma(-m[r], t, T[row * LT + cc]);
        }
      }
    }
    __syncthreads();
  }
  // T[:, J..2J) = X2 = Mi C1 (symmetric up to rounding), T[:, 2J..3J) = X1 = Mi A1, T[:, CH] = Mi h

  // Y = A2 sym(X2);  b12 = b2 + A2 (Mi h);  A12 = A2 X1
  double* O = K.out + ((long)prob * K.nchunk + c) * ELEM;
  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    double acc = 0.0, aacc = 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) {
      const double a = A2m[r * LD + k];
      acc = fma(a, 0.5 * (T[k * LT + J + q] + T[q * LT + J + k]), acc);
      if (!final_) aacc = fma(a, T[k * LT + 2 * J + q], aacc);
    }
    Ym[r * LD + q] = acc;
    if (!final_) O[i] = aacc;
  }
  double b12 = 0.0;
  if (tid < J) {
    b12 = b2v[tid];
#pragma unroll 8
    for (int k = 0; k < J; ++k) b12 = fma(A2m[tid * LD + k], T[k * LT + CH], b12);
  }
  __syncthreads();
  // C12 = C2 + Y A2^T (upper triangle) | b12 -> the next chunk's start state (final) or the level's output element
  double* oC = final_ ? K.starts + ((long)prob * K.nchunk + c + 1) * START : O + J * J + J;
  double* ob = final_ ? oC + SZ : O + J * J;
  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    if (r <= q) {
      double acc = C2[tri(r, q)];
#pragma unroll 8
      for (int k = 0; k < J; ++k) acc = fma(Ym[r * LD + k], A2m[q * LD + k], acc);
      oC[tri(r, q)] = acc;
    }
  }
  if (tid < J) ob[tid] = b12;
  if (final_) return;
  __syncthreads();
  // Y <- Jm2 X1;  eta12 = eta1 + X1^T w;  Jm12 = Jm1 + (A1^T Y + Y^T A1) / 2
  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(J2m[r * LD + k], T[k * LT + 2 * J + q], acc);
    Ym[r * LD + q] = acc;
  }
  if (tid < J) {
    double acc = e1v[tid];
#pragma unroll 8
    for (int k = 0; k < J; ++k) acc = fma(T[k * LT + 2 * J + tid], wv[k], acc);
    O[J * J + J + SZ + tid] = acc;
  }
  __syncthreads();
  double* oJ = O + J * J + J + SZ + J;
  for (int i = tid; i < J * J; i += NT) {
    const int r = i / J, q = i % J;
    if (r <= q) {
      double acc = 0.0;
#pragma unroll 8
      for (int k = 0; k < J; ++k) acc += A1m[k * LD + r] * Ym[k * LD + q] + A1m[k * LD + q] * Ym[k * LD + r];
      oJ[tri(r, q)] = Jm1[tri(r, q)] + 0.5 * acc;
    }
  }
}

template <int J>
void run_levels(const BatchParams& P, hipStream_t s) {
  constexpr int SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ;
  ScanLevel K;
  K.starts = P.starts; K.need_exact = P.need_exact; K.B = P.B; K.nchunk = P.nchunk;
  const double* in = P.elems;
  double* bufs[2] = {P.scan_ws, P.scan_ws + (size_t)P.B * P.nchunk * ELEM};
  int k = 0;
  for (int d = 1; d == 1 || d < P.nchunk - 1; d *= 2, ++k) {
    K.in = in; K.out = bufs[k & 1]; K.d = d;
    const int per = P.nchunk - 1 - d + (d == 1 ? 1 : 0);
    if (per <= 0) break;
    hipLaunchKernelGGL((wide_scan_level_kernel<J>), dim3((unsigned)(P.B * per)), dim3(256), 0, s, K);
    in = bufs[k & 1];
  }
}

}  // namespace

// largest B x nchunk the scan is used for (workgroups per level; profiles/r04zz_wide_midbatch.txt: at width 32 the scan loses
// to the walk from 2048 workgroups per level on, at width 16 it ties at 4096)
int wide_prefix_scan_cap(int width_padded) { return width_padded <= 16 ? 2048 : 1024; }
// most chunks ONE series is cut into (profiles/r04v_single_wide_chunks.txt)
int wide_prefix_scan_max_chunks(int width_padded) { return width_padded <= 16 ? 1024 : 512; }

// doubles of workspace the parallel prefix needs (two level buffers), 0 when this shape keeps the sequential walk
size_t wide_prefix_scan_workspace(int B, int nchunk, int width_padded) {
  long cap = wide_prefix_scan_cap(width_padded);
  if (const char* e = clr::option("CLR_WIDE_SCAN_CAP")) cap = atol(e);  // (tools/gpu_single_wide_chunks2.py)
  if (width_padded > 32) return 0;  // (widths 33..64: the chunks are chained by wide_walk_kernel, wide64_kernels.hip)
  if (nchunk < 8 || (long)B * nchunk > cap) return 0;  // (... and a walk worth cutting)
  const size_t J = width_padded <= 16 ? 16 : 32, SZ = J * (J + 1) / 2;
  return 2 * (size_t)B * nchunk * (J * J + J + SZ + J + SZ);
}

void launch_wide_prefix_scan(const BatchParams& P, int width_padded, hipStream_t s) {
  if (width_padded <= 16) run_levels<16>(P, s);
  else run_levels<32>(P, s);
}

}  // namespace clr
