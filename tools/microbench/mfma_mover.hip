// Can an fp64 MFMA serve as a 64-bit AGPR <-> VGPR mover that runs beside the VALU?
// D = A * I + 0 with A in an AccVGPR and D in a VGPR (or the other way round) copies one
// double per lane in ONE issue slot of the matrix pipe instead of two v_accvgpr_* VALU slots.
// (1) checks the lane mapping (D_lane == A_lane with B = per-block identity),
// (2) times a column visit of the summarize kernel: 8 doubles in, 24 FMAs, 8 doubles out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void mapping(double* out, int variant) {
  const int l = threadIdx.x;
  double a = 100.0 + l;  // value to move
  // candidate identity layouts for B (4x4 per block, one value per lane)
  const int x = l & 3, y = (l >> 2) & 3;
  double ident = (x == y) ? 1.0 : 0.0;
  double d = 0.0;
  double zero = 0.0;
  if (variant == 0)
    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n s_nop 7\n s_nop 7" : "=v"(d) : "v"(a), "v"(ident), "v"(zero));
  else
    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %3\n s_nop 7\n s_nop 7" : "=v"(d) : "v"(a), "v"(ident), "v"(zero));
  out[l] = d;
}

#define ITER 2048
template <int MODE>
__global__ void __launch_bounds__(64) timing(double* out, double seed) {
  const int l = threadIdx.x;
  const int x = l & 3, y = (l >> 2) & 3;
  double ident = (x == y) ? 1.0 : 0.0;
  double v0 = seed + l, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  double m = 1.0000001, c = 1e-9;
  double a0, a1, a2, a3, a4, a5, a6, a7;  // live in AccVGPRs (only touched through "a" constraints)
  asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(((int*)&a0)[0]));
#define MOVE_IN(vd, ad)  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=v"(vd) : "a"(ad), "v"(ident))
#define MOVE_OUT(ad, vs) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=a"(ad) : "v"(vs), "v"(ident))
#define ACC_IN(vd, ad)   asm volatile("v_accvgpr_read_b32 %0, %2\n v_accvgpr_read_b32 %1, %3" : "=v"(((int*)&vd)[0]), "=v"(((int*)&vd)[1]) : "a"(((int*)&ad)[0]), "a"(((int*)&ad)[1]))
#define ACC_OUT(ad, vs)  asm volatile("v_accvgpr_write_b32 %0, %2\n v_accvgpr_write_b32 %1, %3" : "=a"(((int*)&ad)[0]), "=a"(((int*)&ad)[1]) : "v"(((int*)&vs)[0]), "v"(((int*)&vs)[1]))
#define FMA3(v) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c))
  // initialise the accumulators
  if (MODE == 0) { MOVE_OUT(a0, v0); MOVE_OUT(a1, v1); MOVE_OUT(a2, v2); MOVE_OUT(a3, v3); MOVE_OUT(a4, v4); MOVE_OUT(a5, v5); MOVE_OUT(a6, v6); MOVE_OUT(a7, v7); }
  else           { ACC_OUT(a0, v0); ACC_OUT(a1, v1); ACC_OUT(a2, v2); ACC_OUT(a3, v3); ACC_OUT(a4, v4); ACC_OUT(a5, v5); ACC_OUT(a6, v6); ACC_OUT(a7, v7); }
  for (int i = 0; i < ITER; ++i) {
    if (MODE == 0) {
      MOVE_IN(v0, a0); MOVE_IN(v1, a1); MOVE_IN(v2, a2); MOVE_IN(v3, a3); MOVE_IN(v4, a4); MOVE_IN(v5, a5); MOVE_IN(v6, a6); MOVE_IN(v7, a7);
      FMA3(v0); FMA3(v1); FMA3(v2); FMA3(v3); FMA3(v4); FMA3(v5); FMA3(v6); FMA3(v7);
      MOVE_OUT(a0, v0); MOVE_OUT(a1, v1); MOVE_OUT(a2, v2); MOVE_OUT(a3, v3); MOVE_OUT(a4, v4); MOVE_OUT(a5, v5); MOVE_OUT(a6, v6); MOVE_OUT(a7, v7);
    } else if (MODE == 1) {
      ACC_IN(v0, a0); ACC_IN(v1, a1); ACC_IN(v2, a2); ACC_IN(v3, a3); ACC_IN(v4, a4); ACC_IN(v5, a5); ACC_IN(v6, a6); ACC_IN(v7, a7);
      FMA3(v0); FMA3(v1); FMA3(v2); FMA3(v3); FMA3(v4); FMA3(v5); FMA3(v6); FMA3(v7);
      ACC_OUT(a0, v0); ACC_OUT(a1, v1); ACC_OUT(a2, v2); ACC_OUT(a3, v3); ACC_OUT(a4, v4); ACC_OUT(a5, v5); ACC_OUT(a6, v6); ACC_OUT(a7, v7);
    } else {  // interleaved: each move next to arithmetic on another value
      MOVE_IN(v0, a0); MOVE_IN(v1, a1);
      MOVE_IN(v2, a2); FMA3(v0); MOVE_IN(v3, a3); FMA3(v1); MOVE_IN(v4, a4); FMA3(v2); MOVE_IN(v5, a5); FMA3(v3);
      MOVE_IN(v6, a6); FMA3(v4); MOVE_OUT(a0, v0); MOVE_IN(v7, a7); FMA3(v5); MOVE_OUT(a1, v1); FMA3(v6); MOVE_OUT(a2, v2); FMA3(v7);
      MOVE_OUT(a3, v3); MOVE_OUT(a4, v4); MOVE_OUT(a5, v5); MOVE_OUT(a6, v6); MOVE_OUT(a7, v7);
    }
  }
  if (MODE == 1) { ACC_IN(v0, a0); ACC_IN(v7, a7); } else { MOVE_IN(v0, a0); MOVE_IN(v7, a7); }
  out[blockIdx.x * 64 + l] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE>
void run(const char* name, int blocks, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(timing<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(timing<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-46s blocks %4d: %7.3f ms -> %6.1f ns per (8 in, 24 fma, 8 out)  [check %.9g]\n", name, blocks, ms, ms * 1e6 / ITER, h[5]);
}

int main() {
  double* out; hipMalloc(&out, 4096 * 64 * sizeof(double));
  for (int variant = 0; variant < 2; ++variant) {
    hipLaunchKernelGGL(mapping, dim3(1), dim3(64), 0, 0, out, variant);
    double h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1; for (int l = 0; l < 64; ++l) ok &= (h[l] == 100.0 + l);
    printf("mapping variant %d (A=%s): D_lane == A_lane on all lanes: %s   (lanes 0..7: %g %g %g %g %g %g %g %g)\n", variant, variant ? "identity, B=value" : "value, B=identity",
           ok ? "yes" : "NO", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  for (int blocks : {1, 1024}) {
    run<1>("v_accvgpr_read/write (2 + 2 per double)", blocks, out);
    run<0>("mfma movers, grouped", blocks, out);
    run<2>("mfma movers, interleaved with the FMAs", blocks, out);
  }
  return 0;
}
