// celerite_amd/csrc/series_io.hip -- see clr_series_io.h.
//
// What the reference pays for the same hand-over: pybind11's Eigen casters copy every NumPy argument into a fresh Eigen
// object per compute() call (celerite/solver.cpp:467-483: 12 copies, t and diag among them).  Here the series cross the
// PCIe link once per clr_batch_set_series and stay resident in HBM for every later evaluation.
#include "clr_series_io.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace clr {

int staging_create(UploadStaging& s, int device) {
  if (s.ready && s.device == device) return 0;
  staging_destroy(s);
  hipError_t e;
  for (int i = 0; i < UploadStaging::NT; ++i) {
    for (int k = 0; k < 2; ++k) {
      if ((e = hipHostMalloc(reinterpret_cast<void**>(&s.pin[i][k]), UploadStaging::PIECE, hipHostMallocDefault)) != hipSuccess)
        return (int)e;
      if ((e = hipEventCreateWithFlags(&s.ev[i][k], hipEventDisableTiming)) != hipSuccess) return (int)e;
    }
    if ((e = hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking)) != hipSuccess) return (int)e;
  }
  s.device = device;
  s.ready = true;
  return 0;
}

void staging_destroy(UploadStaging& s) {
  for (int i = 0; i < UploadStaging::NT; ++i) {
    for (int k = 0; k < 2; ++k) {
      if (s.pin[i][k]) (void)hipHostFree(s.pin[i][k]);
      if (s.ev[i][k]) (void)hipEventDestroy(s.ev[i][k]);
      s.pin[i][k] = nullptr;
      s.ev[i][k] = nullptr;
    }
    if (s.stream[i]) (void)hipStreamDestroy(s.stream[i]);
    s.stream[i] = nullptr;
  }
  s.ready = false;
  s.device = -1;
}

int upload_parallel(UploadStaging& s, const CopyJob* jobs, int njobs) {
  struct Piece { char* dst; const char* src; size_t bytes; };
  std::vector<Piece> pieces;
  for (int j = 0; j < njobs; ++j) {
    const size_t total = jobs[j].n * sizeof(double);
    for (size_t off = 0; off < total; off += UploadStaging::PIECE)
      pieces.push_back({reinterpret_cast<char*>(jobs[j].dst) + off, reinterpret_cast<const char*>(jobs[j].src) + off,
                        std::min(UploadStaging::PIECE, total - off)});
  }
  if (pieces.empty()) return 0;
  // pieces are handed out dynamically: a thread that lands on a busy core does not hold the others up
  std::atomic<size_t> next(0);
  std::atomic<int> err(0);
  const int nt = (int)std::min<size_t>(UploadStaging::NT, pieces.size());
  auto work = [&](int i) {
    if (hipSetDevice(s.device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
    int used = 0;
    for (;;) {
      const size_t p = next.fetch_add(1);
      if (p >= pieces.size() || err.load()) break;
      const int slot = used & 1;
      hipError_t e = hipSuccess;
      if (used >= 2) e = hipEventSynchronize(s.ev[i][slot]);  // the DMA that last read this buffer is done
      if (e == hipSuccess) {
        memcpy(s.pin[i][slot], pieces[p].src, pieces[p].bytes);
        e = hipMemcpyAsync(pieces[p].dst, s.pin[i][slot], pieces[p].bytes, hipMemcpyHostToDevice, s.stream[i]);
      }
      if (e == hipSuccess) e = hipEventRecord(s.ev[i][slot], s.stream[i]);
      if (e != hipSuccess) { err = (int)e; break; }
      ++used;
    }
    const hipError_t e = hipStreamSynchronize(s.stream[i]);
    if (e != hipSuccess && !err.load()) err = (int)e;
  };
  std::vector<std::thread> th;
  for (int i = 1; i < nt; ++i) th.emplace_back(work, i);
  work(0);
  for (auto& t : th) t.join();
  return err.load();
}

// ---- scans of t on the device ----------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) series_stats_kernel(const double* t, long stride, int N, double* out) {
  const double* tb = t + (long)blockIdx.x * stride;
  double tm = 0.0, dm = 0.0, dmin = INFINITY;
  int nan = 0;
  for (int n = threadIdx.x; n < N; n += 256) {
    const double v = tb[n], a = fabs(v);
    if (v != v) nan = 1;
    tm = fmax(tm, a);
    if (n > 0) {
      const double d = v - tb[n - 1];
      if (d != d) nan = 1;
      dm = fmax(dm, fabs(d));
      dmin = fmin(dmin, d);
    }
  }
  __shared__ double sm[3][256];
  __shared__ int sn[256];
  sm[0][threadIdx.x] = tm; sm[1][threadIdx.x] = dm; sm[2][threadIdx.x] = dmin; sn[threadIdx.x] = nan;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sm[0][threadIdx.x] = fmax(sm[0][threadIdx.x], sm[0][threadIdx.x + w]);
      sm[1][threadIdx.x] = fmax(sm[1][threadIdx.x], sm[1][threadIdx.x + w]);
      sm[2][threadIdx.x] = fmin(sm[2][threadIdx.x], sm[2][threadIdx.x + w]);
      sn[threadIdx.x] |= sn[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = out + (long)blockIdx.x * 4;
    o[0] = sm[0][0]; o[1] = sm[1][0]; o[2] = sm[2][0]; o[3] = sn[0];
  }
}

__global__ void warm_spans_kernel(const double* t, long stride, int nsrc, int wL, int wnchunk, WarmCands cands, double* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nsrc * cands.nk) return;
  const int b = idx / cands.nk, k = idx % cands.nk;
  const double* tb = t + (long)b * stride;
  const long K = cands.K[k];
  double span = INFINITY;
  if (K > wL / 2) span = 0.0;  // (not a usable candidate at this chunk length)
  for (long c = 1; c < wnchunk && span > 0.0; ++c) {
    const long n = c * (long)wL;
    const double d = tb[n] - tb[n - K];
    if (!(d >= span)) span = d;  // (NaN sticks: never eligible)
  }
  out[idx] = span;
}

}  // namespace

void launch_series_stats(const double* t, long stride, int nsrc, int N, double* out, hipStream_t s) {
  hipLaunchKernelGGL(series_stats_kernel, dim3(nsrc), dim3(256), 0, s, t, stride, N, out);
}

void launch_warm_spans(const double* t, long stride, int nsrc, int wL, int wnchunk, WarmCands cands, double* out,
                       hipStream_t s) {
  const int n = nsrc * cands.nk;
  hipLaunchKernelGGL(warm_spans_kernel, dim3((n + 127) / 128), dim3(128), 0, s, t, stride, nsrc, wL, wnchunk, cands, out);
}

}  // namespace clr
