// What clock do the SIMDs run at while EVERY SIMD issues fp64 FMAs back to back, and how many cycles does one
// v_fma_f64 cost there?  (bench.py's measured_valu_fma_rate -- 53.3 TFLOP/s -- against the 78.6
// TFLOP/s of 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz: clock or issue?)
// Every wave times its own stream with s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz), so
//   clock = cycles / realtime x 100 MHz,   cycles per instruction = cycles / instructions issued by the wave.
// hipcc --offload-arch=gfx950 -O3 clock_under_load.hip -o clock_under_load ; ./clock_under_load [waves_per_simd] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x

__global__ void __launch_bounds__(64) load_kernel(double* out, unsigned long long* rec, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
  for (int i = 0; i < iters; ++i) {  // 64 independent-enough FMAs per iteration (8 chains x 8)
    REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                      "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
  if (threadIdx.x == 0) {
    rec[2 * blockIdx.x] = c1 - c0;
    rec[2 * blockIdx.x + 1] = r1 - r0;
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 20000;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int simds = prop.multiProcessorCount * 4;
  for (int waves : {1, simds / 4, simds, simds * wps}) {
    double* out;
    unsigned long long* rec;
    hipMalloc(&out, (size_t)waves * 64 * sizeof(double));
    hipMalloc(&rec, (size_t)waves * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    load_kernel<<<waves, 64>>>(out, rec, iters / 10, 1.0);  // warm-up (clocks ramp)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    load_kernel<<<waves, 64>>>(out, rec, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)waves * 2);
    hipMemcpy(h.data(), rec, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> mhz, cpi;
    for (int w = 0; w < waves; ++w) {
      const double cyc = (double)h[2 * w], real = (double)h[2 * w + 1];
      mhz.push_back(cyc / real * 100.0);
      cpi.push_back(cyc / (64.0 * iters));
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cpi.begin(), cpi.end());
    const double flops = 2.0 * 64 * 64.0 * iters * waves;
    printf("%6d waves (%4.2f per SIMD): kernel %.3f ms = %.1f TFLOP/s fp64 FMA;  shader clock median %.0f MHz (min %.0f, max %.0f);  "
           "shader cycles per v_fma_f64 of ONE wave: median %.2f (min %.2f, max %.2f)\n",
           waves, (double)waves / simds, ms, flops / (ms * 1e-3) / 1e12, mhz[mhz.size() / 2], mhz.front(), mhz.back(),
           cpi[cpi.size() / 2], cpi.front(), cpi.back());
    hipFree(out);
    hipFree(rec);
  }
  return 0;
}
