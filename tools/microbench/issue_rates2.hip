// Round-2 issue-rate table for gfx950 (MI355X): what does a SIMD sustain at 1, 2 and 4
// resident waves, for the instruction mixes the scan kernels are made of?
//   (A) fp64 FMA streams, 32-bit AGPR/VGPR moves, SALU, and mixes of them;
//   (B) fp64 MFMA (4x4x4_4b and 16x16x4) alone and interleaved with INDEPENDENT fp64 FMAs
//       (is the matrix pipe a second fp64 engine beside the VALU, or the same one?);
//   (C) ds_add_f64 (LDS accumulate without return) alone and behind FMAs;
//   (D) the lane layout of v_mfma_f64_4x4x4_4b (A, B, D operands).
// Every kernel runs ITER iterations of an unrolled body; the host reports ns per instruction
// PER SIMD = time / (ITER * instr_per_iter * waves_per_simd) with the whole chip busy
// (1024 * w blocks of one wave).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITER 2048

template <int MODE>
__global__ void __launch_bounds__(64) bench(double* out, double seed) {
  __shared__ double lds[64 * 8];
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
         a6 = a0 + 6, a7 = a0 + 7;
  double m = 1.0000001, c = 1e-9;
  int e0 = threadIdx.x, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
  double q0 = 0, q1 = 0, q2 = 0, q3 = 0;  // MFMA accumulators (VGPR form)
  double ident = ((threadIdx.x & 3) == ((threadIdx.x >> 2) & 3)) ? 1e-3 : 0.0;
  double* my = lds + threadIdx.x;
  for (int k = 0; k < 8; ++k) lds[threadIdx.x + 64 * k] = 0.0;
  for (int i = 0; i < ITER; ++i) {
#define FMA4 asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
#define FMA4B asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#define ACC4 asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3" : "+v"(e0), "+v"(e1), "=v"(e2), "=v"(e3) :: "a0", "a1", "a2", "a3");
#define MOV4 asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %2, %0\n v_mov_b32 %3, %1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
#define SMOV4 asm volatile("s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s20, s21\n s_mov_b32 s22, s23" ::: "s20", "s21", "s22", "s23");
#define MFMA4x4_4 asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(ident), "v"(m));
#define MFMA4x4_1(q) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(q) : "v"(ident), "v"(m));
#define DSADD4 asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %2 offset:512\n ds_add_f64 %0, %3 offset:1024\n ds_add_f64 %0, %4 offset:1536" :: "v"((unsigned)(size_t)my), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "memory");
    if (MODE == 0) { REP8(FMA4 FMA4B) }                                  // 64 fp64 FMA
    else if (MODE == 1) { REP8(ACC4 ACC4) }                              // 64 accvgpr moves
    else if (MODE == 2) { REP8(MOV4 MOV4) }                              // 64 v_mov_b32
    else if (MODE == 3) { REP8(SMOV4 SMOV4) }                            // 64 s_mov_b32
    else if (MODE == 4) { REP8(FMA4 ACC4) }                              // 32 FMA + 32 accvgpr
    else if (MODE == 5) { REP8(FMA4 FMA4B ACC4) }                        // 64 FMA + 32 accvgpr (2:1)
    else if (MODE == 6) { REP8(FMA4 MOV4) }                              // 32 FMA + 32 v_mov
    else if (MODE == 7) { REP8(FMA4 SMOV4) }                             // 32 FMA + 32 s_mov
    else if (MODE == 8) { REP8(FMA4 FMA4B FMA4 ACC4 MOV4 SMOV4) }        // summarize-like: 96 f64 + 32 acc + 32 mov + 32 salu
    else if (MODE == 9) { REP8(MFMA4x4_4 MFMA4x4_4) }                    // 64 MFMA 4x4x4_4b, 4 accumulators
    else if (MODE == 10) { REP8(MFMA4x4_4 FMA4) }                        // 32 MFMA + 32 FMA, independent
    else if (MODE == 11) { REP8(MFMA4x4_1(q0) FMA4 MFMA4x4_1(q1) FMA4B MFMA4x4_1(q2) FMA4 MFMA4x4_1(q3) FMA4B) }  // 32 MFMA + 128 FMA
    else if (MODE == 12) { REP8(DSADD4 DSADD4) }                         // 64 ds_add_f64
    else if (MODE == 13) { REP8(DSADD4 FMA4 FMA4) }                      // 32 ds_add + 64 FMA (a0..a3 chains)
    else if (MODE == 14) { REP8(MFMA4x4_1(q0) FMA4 FMA4B MFMA4x4_1(q1) FMA4 FMA4B) }  // 16 MFMA + 128 FMA
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0 + e1 + e2 + e3 + q0 + q1 + q2 + q3 + lds[threadIdx.x];
}

// 16x16x4 f64 MFMA: 4 doubles of C/D per lane
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(64) bench16(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double m = 1.0000001, c = 1e-9, s = 1e-3;
  double4_t q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  for (int i = 0; i < ITER; ++i) {
#define M16(q) q = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, q, 0, 0, 0);
    if (MODE == 0) { REP8(M16(q0) M16(q1) M16(q2) M16(q3)) }             // 32 MFMA 16x16x4
    else { REP8(M16(q0) FMA4 M16(q1) FMA4 M16(q2) FMA4 M16(q3) FMA4) }   // 32 MFMA + 128 FMA
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + q0[0] + q1[1] + q2[2] + q3[3];
}

// lane layout of v_mfma_f64_4x4x4_4b: A = 1 + lane, B = indicator of lane lb; D[ld] names the A lane
__global__ void layout(double* out) {
  const int l = threadIdx.x;
  for (int lb = 0; lb < 64; ++lb) {
    double a = 1.0 + l, b = (l == lb) ? 1.0 : 0.0, d = 0.0, z = 0.0;
    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n s_nop 7\n s_nop 7" : "=v"(d) : "v"(a), "v"(b), "v"(z));
    out[lb * 64 + l] = d;
  }
}

template <class K>
float time_kernel(K k, int blocks, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 1.0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out; hipMalloc(&out, 8192 * 64 * sizeof(double));
  struct Row { const char* name; int n; void (*k)(double*, double); };
  Row rows[] = {
      {"fp64 FMA x64", 64, bench<0>}, {"accvgpr rd/wr x64", 64, bench<1>}, {"v_mov_b32 x64", 64, bench<2>},
      {"s_mov_b32 x64", 64, bench<3>}, {"FMA 32 + accvgpr 32", 64, bench<4>}, {"FMA 64 + accvgpr 32", 96, bench<5>},
      {"FMA 32 + v_mov 32", 64, bench<6>}, {"FMA 32 + s_mov 32", 64, bench<7>},
      {"summarize-like 96 f64+32 acc+32 mov+32 salu", 192, bench<8>},
      {"MFMA f64 4x4x4_4b x64", 64, bench<9>}, {"MFMA 4x4x4 32 + FMA 32 (indep)", 64, bench<10>},
      {"MFMA 4x4x4 32 + FMA 128 (indep)", 160, bench<11>}, {"ds_add_f64 x64", 64, bench<12>},
      {"ds_add_f64 32 + FMA 64", 96, bench<13>}, {"MFMA 4x4x4 16 + FMA 128 (indep)", 144, bench<14>},
      {"MFMA f64 16x16x4 x32", 32, bench16<0>}, {"MFMA 16x16x4 32 + FMA 128", 160, bench16<1>},
  };
  printf("# ns per instruction per SIMD, whole chip busy (1024*w one-wave blocks); ITER=%d\n", ITER);
  printf("%-46s %10s %10s %10s %10s\n", "mix (instr per iteration)", "1 SIMD", "w=1", "w=2", "w=4");
  for (auto& r : rows) {
    printf("%-46s", r.name);
    {
      float ms = time_kernel(r.k, 1, out);
      printf(" %10.3f", ms * 1e6 / ((double)ITER * r.n));
    }
    for (int w : {1, 2, 4}) {
      float ms = time_kernel(r.k, 1024 * w, out);
      printf(" %10.3f", ms * 1e6 / ((double)ITER * r.n * w));
    }
    printf("\n");
  }
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  hipDeviceSynchronize();
  std::vector<double> h(64 * 64);
  hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  printf("# v_mfma_f64_4x4x4_4b layout: for B-lane lb: list of D-lane<-A-lane pairs\n");
  for (int lb = 0; lb < 64; ++lb) {
    printf("lb %2d:", lb);
    for (int ld = 0; ld < 64; ++ld)
      if (h[lb * 64 + ld] != 0.0) printf(" %d<-%d", ld, (int)h[lb * 64 + ld] - 1);
    printf("\n");
  }
  return 0;
}
