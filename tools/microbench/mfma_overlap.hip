// Round-4 question for BASELINE config 4 (width 32): does v_mfma_f64_16x16x4 on gfx950 run BESIDE the instructions
// a summarize step is made of, or instead of them?  r02a_issue_rates2.txt already shows that fp64 MFMA and fp64 VALU
// FMAs add up (one fp64 datapath).  Here: MFMA interleaved with independent 32-bit moves, DPP moves, SALU, LDS reads
// (the ~190 non-fp64 instructions of a wide summarize step), and MFMA waves sharing a SIMD with FMA waves.
// Output: ns per loop iteration per SIMD-resident wave set, whole chip busy; compare a mix with the sum / the max of its
// parts.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define ITER 1024
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) bench(double* out, double seed, int split) {
  __shared__ double lds[64 * 8];
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double m = 1.0000001, c = 1e-9, s = 1e-3;
  int e0 = threadIdx.x, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
  double4_t q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  double r0 = 0, r1 = 0;
  for (int k = 0; k < 8; ++k) lds[threadIdx.x + 64 * k] = k;
  const unsigned my = (unsigned)(size_t)(lds + threadIdx.x);
  // role of this wave in the mixed-wave modes: by block parity (split = 1) or by half of the grid (split = 2)
  const bool role = split == 1 ? (blockIdx.x & 1) : (blockIdx.x >= gridDim.x / 2);
  for (int i = 0; i < ITER; ++i) {
#define FMA4 asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
#define MOV4 asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %2, %0\n v_mov_b32 %3, %1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
#define DPP4 asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
#define SMOV4 asm volatile("s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s20, s21\n s_mov_b32 s22, s23" ::: "s20", "s21", "s22", "s23");
#define LDS4 asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n ds_read_b64 %0, %2 offset:1024\n ds_read_b64 %1, %2 offset:1536\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(my) : "memory");
#define M16(q) q = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, q, 0, 0, 0);
    if (MODE == 0) { REP8(M16(q0) M16(q1) M16(q2) M16(q3)) }                                        // 32 MFMA
    else if (MODE == 1) { REP8(MOV4 MOV4 MOV4 MOV4 MOV4 MOV4 MOV4 MOV4) }                            // 256 v_mov
    else if (MODE == 2) { REP8(M16(q0) MOV4 MOV4 M16(q1) MOV4 MOV4 M16(q2) MOV4 MOV4 M16(q3) MOV4 MOV4) }  // 32 MFMA + 256 v_mov
    else if (MODE == 3) { REP8(DPP4 DPP4 DPP4 DPP4 DPP4 DPP4 DPP4 DPP4) }                            // 256 dpp mov
    else if (MODE == 4) { REP8(M16(q0) DPP4 DPP4 M16(q1) DPP4 DPP4 M16(q2) DPP4 DPP4 M16(q3) DPP4 DPP4) }  // 32 MFMA + 256 dpp
    else if (MODE == 5) { REP8(SMOV4 SMOV4 SMOV4 SMOV4 SMOV4 SMOV4 SMOV4 SMOV4) }                    // 256 s_mov
    else if (MODE == 6) { REP8(M16(q0) SMOV4 SMOV4 M16(q1) SMOV4 SMOV4 M16(q2) SMOV4 SMOV4 M16(q3) SMOV4 SMOV4) }  // 32 MFMA + 256 s_mov
    else if (MODE == 7) { REP8(LDS4 LDS4 LDS4 LDS4) }                                                // 128 ds_read_b64
    else if (MODE == 8) { REP8(M16(q0) LDS4 M16(q1) LDS4 M16(q2) LDS4 M16(q3) LDS4) }                // 32 MFMA + 128 ds_read_b64
    else if (MODE == 9) { REP8(FMA4 FMA4 FMA4 FMA4) }                                                // 128 FMA
    else if (MODE == 10) { REP8(M16(q0) FMA4 M16(q1) FMA4 M16(q2) FMA4 M16(q3) FMA4) }               // 32 MFMA + 128 FMA, one wave
    else if (MODE == 11) {                                                                           // MFMA waves beside FMA waves
      if (role) { REP8(FMA4 FMA4 FMA4 FMA4) } else { REP8(M16(q0) M16(q1) M16(q2) M16(q3)) }
    } else if (MODE == 12) {                                                                         // MFMA waves beside v_mov waves
      if (role) { REP8(MOV4 MOV4 MOV4 MOV4 MOV4 MOV4 MOV4 MOV4) } else { REP8(M16(q0) M16(q1) M16(q2) M16(q3)) }
    } else if (MODE == 13) {                                                                         // summarize-like mix beside MFMA
      if (role) { REP8(FMA4 FMA4 MOV4 DPP4 SMOV4 LDS4) } else { REP8(M16(q0) M16(q1) M16(q2) M16(q3)) }
    } else if (MODE == 14) { REP8(FMA4 FMA4 MOV4 DPP4 SMOV4 LDS4) }                                  // that mix alone: 64 FMA + 128 other
    else if (MODE == 15) { REP8(M16(q0) FMA4 MOV4 M16(q1) FMA4 DPP4 M16(q2) SMOV4 M16(q3) LDS4) }    // mix + 32 MFMA in one wave
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + e0 + e1 + e2 + e3 + q0[0] + q1[1] + q2[2] + q3[3] + r0 + r1;
}

template <class K>
float time_kernel(K k, int blocks, double* out, int split) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 1.0, split);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 1.0, split);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out; hipMalloc(&out, 8192 * 64 * sizeof(double));
  struct Row { const char* name; void (*k)(double*, double, int); };
  Row rows[] = {
      {"32 MFMA f64 16x16x4", bench<0>}, {"256 v_mov_b32", bench<1>}, {"32 MFMA + 256 v_mov (one wave)", bench<2>},
      {"256 v_mov_b32_dpp", bench<3>}, {"32 MFMA + 256 dpp (one wave)", bench<4>},
      {"256 s_mov_b32", bench<5>}, {"32 MFMA + 256 s_mov (one wave)", bench<6>},
      {"128 ds_read_b64 (waitcnt per 4)", bench<7>}, {"32 MFMA + 128 ds_read (one wave)", bench<8>},
      {"128 fp64 FMA", bench<9>}, {"32 MFMA + 128 FMA (one wave)", bench<10>},
      {"MFMA waves | FMA waves (128)", bench<11>}, {"MFMA waves | v_mov waves (256)", bench<12>},
      {"MFMA waves | mix waves (64 FMA+128 other)", bench<13>}, {"mix alone: 64 FMA + 32 mov/dpp/smov/lds", bench<14>},
      {"mix + 32 MFMA (one wave)", bench<15>},
  };
  printf("# ns per loop iteration and wave-per-SIMD (time / (ITER * w)), whole chip busy: 1024*w one-wave blocks; ITER=%d\n", ITER);
  printf("# mixed-wave rows: half of the waves run the MFMA stream, half the other stream (w=2p: split by block parity; w=2h: by grid half)\n");
  printf("%-46s %10s %10s %10s %10s %10s\n", "loop body", "w=1", "w=2", "w=4", "w=2p", "w=2h");
  for (auto& r : rows) {
    printf("%-46s", r.name);
    for (int w : {1, 2, 4}) {
      float ms = time_kernel(r.k, 1024 * w, out, 0);
      printf(" %10.1f", ms * 1e6 / ((double)ITER * w));
    }
    for (int split : {1, 2}) {
      float ms = time_kernel(r.k, 2048, out, split);
      printf(" %10.1f", ms * 1e6 / ((double)ITER * 2));
    }
    printf("\n");
  }
  return 0;
}
