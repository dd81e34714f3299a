// Issue cost of the fp64 VALU instructions the scan kernels are made of, for a LONE
// wave per SIMD (the configuration of summarize / replay): cycles per instruction of
// independent streams and of dependent chains.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 4096

template <int MODE>
__global__ void __launch_bounds__(64) bench(double* out, long long* cyc, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
         a6 = a0 + 6, a7 = a0 + 7;
  double m = 1.0000001, c = 1e-9;
  int e = 1;
  const unsigned long long mask = __builtin_amdgcn_read_exec() >> 1;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i) {
    if (MODE == 0) {  // 8 independent fma chains, 16 instr per iteration
      REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a4), "+v"(a5) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (MODE == 1) {  // one dependent fma chain
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
    } else if (MODE == 2) {  // two interleaved chains
      REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                         "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
    } else if (MODE == 3) {  // three interleaved chains (9 per group -> count 9)
      REP16(asm volatile("v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %1, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %0, %0, %3, %4\n"
                         "v_fma_f64 %1, %1, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %1, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(m), "v"(c));)
    } else if (MODE == 4) {  // independent mul
      REP16(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(m));
            asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a2), "+v"(a3) : "v"(m));
            asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a4), "+v"(a5) : "v"(m));
            asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (MODE == 5) {  // independent add
      REP16(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a2), "+v"(a3) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a4), "+v"(a5) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 6) {  // accvgpr round trips (32-bit moves), independent
      REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                         "v_accvgpr_read_b32 %0, a4\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a6\n v_accvgpr_read_b32 %3, a7"
                         : "+v"(e), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");)
    } else if (MODE == 7) {  // v_rcp_f64 independent
      REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 8) {  // v_rndne_f64
      REP16(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 9) {  // v_ldexp_f64
      REP16(asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));)
    } else if (MODE == 10) {  // v_frexp_mant_f64
      REP16(asm volatile("v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3\n v_frexp_mant_f64 %4, %4\n v_frexp_mant_f64 %5, %5\n v_frexp_mant_f64 %6, %6\n v_frexp_mant_f64 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 11) {  // v_cndmask_b32 (vcc), independent
      REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                         : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) : "v"(e) : "vcc");)
    } else if (MODE == 12) {  // fma with an SGPR-pair operand (constant from scalar regs)
      REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(m), "s"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a2), "+v"(a3) : "v"(m), "s"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a4), "+v"(a5) : "v"(m), "s"(c));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a6), "+v"(a7) : "v"(m), "s"(c));)
    } else if (MODE == 13) {  // v_cvt_i32_f64 + v_cmp_gt_f64
      REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0\n"
                         "v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
    } else if (MODE == 14) {  // v_mov_b64 (VGPR copies)
      REP16(asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n v_mov_b64 %0, %5\n v_mov_b64 %1, %5\n v_mov_b64 %2, %5\n v_mov_b64 %3, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
    } else if (MODE == 16) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3)
      REP16(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5\n"
                         "v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5"
                         : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) : "v"(e), "s"(mask));)
    } else if (MODE == 17) {  // v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n v_mov_b32 %0, %5\n v_mov_b32 %1, %5\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5"
                         : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) : "v"(e), "v"(((int*)&a5)[0]));)
    } else if (MODE == 18) {  // v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                         : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) : "v"(e));)
    } else if (MODE == 19) {  // v_readlane_b32 -> sgpr
      REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                         "v_readlane_b32 s20, %0, 11\n v_readlane_b32 s21, %1, 13\n v_readlane_b32 s22, %2, 15\n v_readlane_b32 s23, %3, 17"
                         :: "v"(((int*)&a0)[0]), "v"(((int*)&a1)[0]), "v"(((int*)&a2)[0]), "v"(((int*)&a3)[0]) : "s20", "s21", "s22", "s23");)
    } else if (MODE == 20) {  // s_mov_b32 (SALU)
      REP16(asm volatile("s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s20, s21\n s_mov_b32 s22, s23"
                         ::: "s20", "s21", "s22", "s23");)
    } else if (MODE == 21) {  // fma interleaved 1:1 with s_mov (does SALU issue beside VALU?)
      REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mov_b32 s20, s21\n v_fma_f64 %1, %1, %2, %3\n s_mov_b32 s22, s23\n v_fma_f64 %0, %0, %2, %3\n s_mov_b32 s20, s21\n v_fma_f64 %1, %1, %2, %3\n s_mov_b32 s22, s23"
                         : "+v"(a0), "+v"(a1) : "v"(m), "v"(c) : "s20", "s21", "s22", "s23");)
    } else if (MODE == 22) {  // v_cndmask_b64-style select: two cndmask per double, mask in vcc, preceded by v_cmp
      REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_fma_f64 %0, %0, %5, %6\n"
                         "v_cmp_gt_f64 vcc, %1, %0\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_fma_f64 %1, %1, %5, %6"
                         : "+v"(a0), "+v"(a1), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]) : "v"(e), "v"(m), "v"(c) : "vcc");)
    } else if (MODE == 15) {  // dependent chain of 4 with 3 independent fillers between (distance 4)
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                         "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int blocks, double* out, long long* cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  const double n = (double)ITER * per_iter;
  // readcyclecounter ticks at a fixed 100 MHz: use wall time and report ns per
  // instruction; cycles follow from the clock (printed by the caller's reference row)
  printf("%-44s blocks %5d: %8.3f ms  %7.3f ns/instr  (counter ticks/instr %.4f)\n", name, blocks, ms,
         ms * 1e6 / n, (double)h[0] / n);
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 4096 * 64 * sizeof(double));
  hipMalloc(&cyc, 4096 * sizeof(long long));
  for (int blocks : {1, 256, 512, 1024, 2048}) {
    run<0>("fma_f64, 8 independent chains", 128, blocks, out, cyc);
    run<1>("fma_f64, 1 dependent chain", 128, blocks, out, cyc);
    run<2>("fma_f64, 2 interleaved chains", 128, blocks, out, cyc);
    run<3>("fma_f64, 3 interleaved chains", 128, blocks, out, cyc);
    run<15>("fma_f64, 4 interleaved chains", 128, blocks, out, cyc);
    run<4>("mul_f64 independent", 128, blocks, out, cyc);
    run<5>("add_f64 independent", 128, blocks, out, cyc);
    run<12>("fma_f64 with SGPR operand", 128, blocks, out, cyc);
    run<6>("accvgpr write/read b32", 128, blocks, out, cyc);
    run<14>("v_mov_b64", 128, blocks, out, cyc);
    run<11>("v_cndmask_b32", 128, blocks, out, cyc);
    run<13>("v_cmp_gt_f64", 128, blocks, out, cyc);
    run<16>("v_cndmask_b32, SGPR-pair mask", 128, blocks, out, cyc);
    run<22>("cmp + 2 cndmask + fma mix", 128, blocks, out, cyc);
    run<17>("v_mov_b32", 128, blocks, out, cyc);
    run<18>("v_add_u32", 128, blocks, out, cyc);
    run<19>("v_readlane_b32", 128, blocks, out, cyc);
    run<20>("s_mov_b32", 128, blocks, out, cyc);
    run<21>("fma_f64 + s_mov 1:1 (per pair)", 128, blocks, out, cyc);
    run<7>("v_rcp_f64", 128, blocks, out, cyc);
    run<8>("v_rndne_f64", 128, blocks, out, cyc);
    run<9>("v_ldexp_f64", 128, blocks, out, cyc);
    run<10>("v_frexp_mant_f64", 128, blocks, out, cyc);
  }
  return 0;
}
