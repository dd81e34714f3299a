// celerite_amd/csrc/wide_scan32.hip -- the lazy summarize flavours of the wide scan at the padded width 32 (BASELINE
// configs[4]'s dominant kernel: wide_scan_kernel<32, ., 1, true, false, PAIRED, GAPS>) as a translation unit of their own:
// wide_kernels.hip compiled a second time with CLR_WIDE_SCAN32_ONLY, under -mllvm -amdgpu-sched-strategy=max-ilp (Makefile).
// The strategy is worth 3.5 % on this kernel (9.52 -> 9.20 ms for 256 x 1e5 x width 32, same bits) and costs the other
// kernels of wide_kernels.hip 8-30 % (profiles/r06zj_wide_sched_ab.txt, r06zl_wide_ilp_other_widths.txt).
#define CLR_WIDE_SCAN32_ONLY 1
#include "wide_kernels.hip"
