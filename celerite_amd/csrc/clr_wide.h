// celerite_amd/csrc/clr_wide.h -- device helpers shared by the wave-per-problem kernels
// (wide_kernels.hip: batched log-likelihood for widths 9..64; grad_kernels.hip:
// forward-mode gradient): lane geometry, DPP butterflies, wave-uniform lane reads.
#pragma once
#include <hip/hip_runtime.h>

#include "clr_core.h"

namespace clr {

template <int WMAX>
struct WideGeom {
  static constexpr int LPR = 64 / WMAX;     // lanes per row
  static constexpr int COLS = WMAX / LPR;   // columns per lane
};

// lane `k` (wave-uniform) of a per-lane double: two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// v + (v of the DPP-selected lane): two v_mov_b32_dpp + one v_add_f64, no LDS crossbar
// latency (a __shfl_xor is two ds_bpermute_b32, ~100 cycles each for a lone wave).
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  // (mov_dpp: every lane has a valid source for these controls, no `old` value to set up)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // row_half_mirror: i <-> 7 - i within 8 lanes
constexpr int DPP_MIRROR = 0x140;       // row_mirror:      i <-> 15 - i within 16 lanes

// Sum over the rows of the matrix of a value every lane of a row holds identically.
// Rows are LPR adjacent lanes, so the first log2(LPR) butterfly stages are skipped; the
// four 16-lane groups are combined through SGPRs.  The result is wave-uniform.
template <int LPR>
__device__ __forceinline__ double row_sum(double v) {
  if (LPR < 2) v = dpp_add<DPP_QUAD_XOR1>(v);
  if (LPR < 4) v = dpp_add<DPP_QUAD_XOR2>(v);
  v = dpp_add<DPP_HALF_MIRROR>(v);
  v = dpp_add<DPP_MIRROR>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

constexpr int DPP_ROW_ROR4 = 0x124;     // row_ror:4: lane i <- lane (i + 4) mod 16 of its 16-lane row
constexpr int DPP_ROW_ROR8 = 0x128;     // row_ror:8

// Two such sums with ONE butterfly tree (rows of LPR >= 2 lanes): the first lane of every row brings the first
// sum's term, the second lane the second sum's (the caller passes seg == 0 ? a : b); further lanes of a row bring
// nothing.  Every stage (xor 2, rotate by 4, rotate by 8) keeps the lane parity, so even lanes end with sum a and odd
// lanes with sum b of their 16-lane group.
template <int LPR>
__device__ __forceinline__ void row_sum2(double mine, int seg, double* sa, double* sb) {
  static_assert(LPR >= 2, "needs two lanes per row");
  double v = (LPR > 2 && seg >= 2) ? 0.0 : mine;
  v = dpp_add<DPP_QUAD_XOR2>(v);
  v = dpp_add<DPP_ROW_ROR4>(v);
  v = dpp_add<DPP_ROW_ROR8>(v);
  *sa = (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
  *sb = (lane_value(v, 1) + lane_value(v, 17)) + (lane_value(v, 33) + lane_value(v, 49));
}

// v + the value of the lane 16 (32) lanes away: gfx950's v_permlane16_swap / v_permlane32_swap exchange odd and even
// 16-lane rows (the two 32-lane halves) of two registers -- with both operands copies of v, the two results are v of
// the own row and v of the partner row in EVERY lane, lane positions inside the rows unchanged.  No LDS, no SGPR trip.
__device__ __forceinline__ double swap_add16(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_add32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_EVEN = 0xA0;  // quad_perm:[0,0,2,2]: both lanes of a pair read the even one
constexpr int DPP_QUAD_ODD = 0xF5;   // quad_perm:[1,1,3,3]

// row_sum with the total delivered to every lane as a vector value: the in-row butterfly, then the permlane swaps
template <int LPR>
__device__ __forceinline__ double row_sum_all(double v) {
  if (LPR < 2) v = dpp_add<DPP_QUAD_XOR1>(v);
  if (LPR < 4) v = dpp_add<DPP_QUAD_XOR2>(v);
  v = dpp_add<DPP_HALF_MIRROR>(v);
  v = dpp_add<DPP_MIRROR>(v);
  return swap_add32(swap_add16(v));
}

// row_sum2 with both sums delivered to EVERY lane as vector values (no v_readlane / SGPR round trip): the butterfly is
// completed across the four 16-lane rows with the permlane swaps, then the even lane's total (sum a) and the odd
// lane's (sum b) are copied to their pair.  The consumers (D, 1 / D, x ...) are per-lane fp64 operations anyway.
template <int LPR>
__device__ __forceinline__ void row_sum2_all(double mine, int seg, double* sa, double* sb) {
  static_assert(LPR >= 2, "needs two lanes per row");
  double v = (LPR > 2 && seg >= 2) ? 0.0 : mine;
  v = dpp_add<DPP_QUAD_XOR2>(v);
  v = dpp_add<DPP_ROW_ROR4>(v);
  v = dpp_add<DPP_ROW_ROR8>(v);
  v = swap_add16(v);
  v = swap_add32(v);
  *sa = dpp_mov<DPP_QUAD_EVEN>(v);
  *sb = dpp_mov<DPP_QUAD_ODD>(v);
}

}  // namespace clr
