// What can the chip WRITE?  The materialising replay stores 20.5 GB of factor per step (1024 x 1e5 x width 8, the
// reference's four arrays) and reads 2.5 GB of series; it takes 3.8 - 5.1 ms depending on the box, i.e. 4.5 - 6 TB/s.
// This file measures the ceiling that number should be held against, with no arithmetic at all:
//   fill      : every thread stores 8-byte values, consecutive lanes consecutive addresses, grid-stride (plain / nontemporal)
//   fill16    : the same with 16-byte stores
//   replay    : the replay's own pattern -- a wave owns 64 chunks of one problem and stores, step after step, `rows`
//               rows of 512 B that are `nchunk * 8` bytes apart (rows = 25: D, W, u, phi of width 8; 9: the lean layout)
//   copy      : read + write of half the bytes each (what relayout_kernel does)
// hipcc --offload-arch=gfx950 -O3 store_ceiling.hip -o store_ceiling ; ./store_ceiling [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool NT>
__global__ void __launch_bounds__(256) fill_kernel(double* __restrict__ p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}

template <bool NT>
__global__ void __launch_bounds__(256) fill16_kernel(double2* __restrict__ p, size_t n, double v) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  v2 x = {v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<v2*>(p) + i); else reinterpret_cast<v2*>(p)[i] = x;
  }
}

// grid: (nchunk / 64 waves per problem) x problems, one wave per block of 64 threads (as the replay: waves_per_eu small)
template <bool NT>
__global__ void __launch_bounds__(64) replay_pattern_kernel(double* __restrict__ p, int L, int nchunk, int rows, double v) {
  const int wpp = nchunk / 64;
  const int b = blockIdx.x / wpp, c = (blockIdx.x % wpp) * 64 + threadIdx.x;
  double* base = p + (size_t)b * rows * L * nchunk + c;
  for (int i = 0; i < L; ++i) {
    for (int j = 0; j < rows; ++j) {
      double* q = base + ((size_t)i * rows + j) * nchunk;
      if (NT) __builtin_nontemporal_store(v + j, q); else *q = v + j;
    }
  }
}

__global__ void __launch_bounds__(256) copy_kernel(const double* __restrict__ s, double* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}

template <class F>
static double timed(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int B = 1024, N = 100000, nchunk = 256, L = (N + nchunk - 1) / nchunk;
  const int rows_ref = 25, rows_lean = 9;
  const size_t n = (size_t)B * rows_ref * L * nchunk;  // doubles (20.6 GB)
  double* p;
  if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const double gb = n * 8.0 / 1e9;
  const int reps = 5;
  for (int grid : {2048, 8192, 65536}) {
    double ms = timed([&] { fill_kernel<false><<<grid, 256>>>(p, n, 1.0); }, reps);
    printf("fill   8 B plain        grid %6d: %.3f ms  %.2f TB/s\n", grid, ms, gb / ms);
    ms = timed([&] { fill_kernel<true><<<grid, 256>>>(p, n, 1.0); }, reps);
    printf("fill   8 B nontemporal  grid %6d: %.3f ms  %.2f TB/s\n", grid, ms, gb / ms);
    ms = timed([&] { fill16_kernel<false><<<grid, 256>>>((double2*)p, n / 2, 1.0); }, reps);
    printf("fill  16 B plain        grid %6d: %.3f ms  %.2f TB/s\n", grid, ms, gb / ms);
    ms = timed([&] { fill16_kernel<true><<<grid, 256>>>((double2*)p, n / 2, 1.0); }, reps);
    printf("fill  16 B nontemporal  grid %6d: %.3f ms  %.2f TB/s\n", grid, ms, gb / ms);
  }
  {
    double ms = timed([&] { hipMemsetAsync(p, 0, n * 8, 0); }, reps);
    printf("hipMemsetAsync                      : %.3f ms  %.2f TB/s\n", ms, gb / ms);
  }
  for (int rows : {rows_ref, rows_lean}) {
    const double g = (double)B * rows * L * nchunk * 8.0 / 1e9;
    double ms = timed([&] { replay_pattern_kernel<false><<<B * (nchunk / 64), 64>>>(p, L, nchunk, rows, 1.0); }, reps);
    printf("replay pattern %2d rows plain       : %.3f ms  %.2f TB/s (%.1f GB)\n", rows, ms, g / ms, g);
    ms = timed([&] { replay_pattern_kernel<true><<<B * (nchunk / 64), 64>>>(p, L, nchunk, rows, 1.0); }, reps);
    printf("replay pattern %2d rows nontemporal : %.3f ms  %.2f TB/s (%.1f GB)\n", rows, ms, g / ms, g);
  }
  {
    double ms = timed([&] { copy_kernel<<<8192, 256>>>(p, p + n / 2, n / 2); }, reps);
    printf("copy (read half, write half)        : %.3f ms  %.2f TB/s (read + write)\n", ms, gb / ms);
  }
  hipFree(p);
  return 0;
}
