// celerite_amd/csrc/clr_series_io.h -- host -> HBM transfer of a batch's series (t, diag, y: 2.46 GB at the headline
// shape) and the scans of t the plan needs, on the device.  Implementation: series_io.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace clr {

// Pinned staging for uploads from PAGEABLE host memory (what a NumPy array is): NT host threads, each with two pinned
// buffers and its own stream.  A thread copies a piece into one buffer (a host memcpy, ~10 GB/s per thread), queues the
// DMA of that buffer and meanwhile fills the other one; the threads' DMAs share the PCIe link.  One hipMemcpy from
// pageable memory does the same staging on ONE thread: 10-12 GB/s (profiles/r03end_bench.json: 196 ms for 2.46 GB).
struct UploadStaging {
  static constexpr int NT = 8;
  static constexpr size_t PIECE = (size_t)4 << 20;  // bytes per piece
  double* pin[NT][2] = {};
  hipStream_t stream[NT] = {};
  hipEvent_t ev[NT][2] = {};
  int device = -1;
  bool ready = false;
};
struct CopyJob {
  double* dst;        // device
  const double* src;  // host (pageable or pinned)
  size_t n;           // doubles
};
int staging_create(UploadStaging& s, int device);  // 0 on success, else a hipError_t
void staging_destroy(UploadStaging& s);
// copies all jobs, returns when every byte has arrived (0 on success, else a hipError_t)
int upload_parallel(UploadStaging& s, const CopyJob* jobs, int njobs);

// per source series (nsrc = 1 for a shared series): max |t|, largest |step|, smallest step (negative: not sorted),
// NaN seen (0 / 1) -> out[nsrc][4]
void launch_series_stats(const double* t, long stride, int nsrc, int N, double* out, hipStream_t s);
// per source series and candidate k: the shortest time the cand[k] samples in front of a warm-path chunk boundary
// (samples c wL, c = 1 .. wnchunk - 1) span; 0 for candidates above wL / 2 -> out[nsrc][nk]
struct WarmCands { int K[16]; int nk; };
void launch_warm_spans(const double* t, long stride, int nsrc, int wL, int wnchunk, WarmCands cands, double* out,
                       hipStream_t s);

}  // namespace clr
