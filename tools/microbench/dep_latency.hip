// Dependent-issue latency of fp64 FMA on gfx950 (MI355X): ns per instruction per WAVE when a wave's FMAs
// form C independent chains (C = 1: every FMA waits for the previous one), at 1 and 2 resident waves per SIMD.
// issue_rates2.hip always used 8 chains, i.e. measured throughput only.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
#define REP16(x) x x x x x x x x x x x x x x x x

template <int C>
__global__ void __launch_bounds__(64) chains(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < ITER; ++i) {
    if (C == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));) }
    if (C == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));) }
    if (C == 4) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));) }
    if (C == 8) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c)); asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));) }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int C>
double run(int waves_per_simd, double* d) {
  const int blocks = 1024 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<C>, dim3(blocks), dim3(64), 0, 0, d, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<C>, dim3(blocks), dim3(64), 0, 0, d, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_iter = (C == 8) ? 128.0 : 64.0;
  return ms * 1e6 / (ITER * per_iter);  // ns per instruction per WAVE (not per SIMD)
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * 4096);
  printf("# fp64 FMA, ns per instruction per WAVE; C independent chains per wave; whole chip busy\n");
  printf("chains   1 wave/SIMD   2 waves/SIMD   4 waves/SIMD\n");
  printf("%6d %13.3f %14.3f %14.3f\n", 1, run<1>(1, d), run<1>(2, d), run<1>(4, d));
  printf("%6d %13.3f %14.3f %14.3f\n", 2, run<2>(1, d), run<2>(2, d), run<2>(4, d));
  printf("%6d %13.3f %14.3f %14.3f\n", 4, run<4>(1, d), run<4>(2, d), run<4>(4, d));
  printf("%6d %13.3f %14.3f %14.3f\n", 8, run<8>(1, d), run<8>(2, d), run<8>(4, d));
  return 0;
}
