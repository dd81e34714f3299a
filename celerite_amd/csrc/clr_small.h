// celerite_amd/csrc/clr_small.h -- one short series factorised in one launch (small_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace clr {

struct SmallParams {
  int N, L;                 // samples, samples per lane (chunk)
  double coeff[24];         // a_real c_real a_comp b_comp c_comp d_comp, packed (kernel arguments: no upload)
  double jitter;
  const double *t, *diag, *y;   // y may be null (no right-hand side announced)
  double *phi, *u, *W, *D;      // the factor, reference storage (cholesky.h:703-706)
  double* out;                  // [4]: status (0 settled, -1 take the general route), log det, quadratic form, residual
  double max_residual;          // largest relative end-state / start-state mismatch that counts as consistent
};

struct BatchParams;
// a batch of short problems (widths 1..4, 512 <= N <= 32768), one workgroup of `threads` (64, 128 or 256) lanes per
// problem, one launch; needs P.need_scan (problems it could not certify are left pending for the scan pipeline)
bool small_batch_supported(int JR, int JC, int N);
bool launch_small_batch(int JR, int JC, const BatchParams& P, int threads, hipStream_t s);

bool small_compute_supported(int JR, int JC, int N);
// threads: 64, 128 or 256 (a power of two), with threads * L >= N
bool launch_small_compute(int JR, int JC, const SmallParams& P, int threads, bool fast, hipStream_t s);

}  // namespace clr
