// celerite_amd/csrc/clr_split_kernels.h
//
// summarize for the widest register-resident shapes (J = 7, 8), split into two ROLES that
// run as two different waves on the same SIMD.
//
// Why.  The single-wave summarize (clr_core.h: summarize_chunk) carries the whole transfer
// element (A, b, C, eta, Jm) = 152 doubles per lane at J = 8 -- 304 registers of state
// alone, 425 with temporaries -- so it runs ONE wave per SIMD with ~170 registers of it
// parked in AGPRs (300 v_accvgpr moves per step, a quarter of its issue slots).  Measured
// on MI355X (tools/microbench/issue_rates2.hip, profiles/r02a_issue_rates2.txt): a lone
// wave issues one instruction per ~2.1-2.6 ns whatever it is, while two resident waves
// overlap each other's 32-bit moves, LDS and scalar instructions behind the fp64 FMAs
// (the same instruction mix runs 1.44x faster at two waves per SIMD).  The element does
// not fit twice into the 512 registers of a SIMD -- but its update has a one-way data
// flow (clr_core.h header):
//
//   TRAJECTORY  (C, b) and the zero-start sums: the reference recurrence itself from the
//               zero state (cholesky.h:154-178, :350-356).  It produces, per step,
//               u, Phi, pw = Phi W, 1/D, x/D.
//   RIDERS      column j of A needs only those per-step vectors:
//                   r_j = u . A[:, j] ;  A[:, j] <- Phi A[:, j] - pw r_j
//               and  eta_j -= r_j x/D ,  Jm[k][j] -= r_k r_j / D  (k <= j).
//
// So wave T ("trajectory") owns C and b (185 registers) and publishes the per-step vectors
// through one LDS slot; wave R ("riders") owns A and eta in registers and accumulates Jm
// -- which is never read inside the loop -- in LDS (read-modify-write of 18 double2 per
// step, hidden behind the other wave's FMAs).  Both fit in 256 registers WITHOUT spills (a
// scratch reload in T would queue behind its series prefetch: vmcnt is in order), so a SIMD
// holds one T and one R wave; a workgroup is 8 waves = 4 (problem, 64-chunk block) sets, one
// per SIMD, and each wave picks its role from the SIMD it was actually placed on (HW_ID), so
// every SIMD gets exactly one T and one R whatever the dispatch order.  The balance between
// the roles does not matter (they share one issue port); the total instruction count does.
// LDS is then full (slots 4 x 2 x 10.8 KB + Jm 4 x 18.4 KB = 160 KB), so T reads the series from the
// chunk-interleaved copy [problem][i][chunk] (one coalesced 512-B load per array and step,
// relayout_kernel in api.hip, made once per set_series) instead of staging tiles.
//
// Synchronisation: R runs one step behind T, through TWO slots and one workgroup barrier per
// step: T fills slot i & 1 during step i and then meets R at barrier B(i); R, after B(i), reads
// slot i & 1 into registers and folds the step in while T computes step i + 1 into the other
// slot.  T may overwrite slot i & 1 again in step i + 2, i.e. after B(i + 1), which R only
// reaches once its reads of step i are done.  (Measured alternatives: two barriers around one
// slot 3.40 ms; a barrier-free hand-over through a marker in the slot, R polling with s_sleep,
// 3.31 ms; this 3.20 ms; without any synchronisation -- wrong results -- 2.94 ms.  The rest of
// the gap to the single-wave kernel's 3.5 ms is R idling: its step is shorter than T's.)
#pragma once

#include "clr_batch_kernels.h"

namespace clr {

// fields of the published step: u of the complex rows (a real row's u is the wave-uniform
// a_real), distinct phi, pw, 1/D, x/D; field f of a lane at slot[f * 64]
template <int JR, int JC>
struct SplitLink {
  static constexpr int J = JR + 2 * JC;
  static constexpr int NU = 2 * JC;
  static constexpr int F_U = 0, F_PHI = NU, F_PW = NU + JR + JC, F_INVD = F_PW + J, F_XS = F_INVD + 1;
  static constexpr int NPAY = F_XS + 1;
  static constexpr int SZ = J * (J + 1) / 2;
  static constexpr int NJM = (SZ + 1) / 2;  // double2 cells of the packed Jm per lane
};

// packed upper-triangle index e -> (k, j), k <= j  (inverse of tri())
CLR_HD constexpr int tri_col(int e) { int j = 0; while ((j + 1) * (j + 2) / 2 <= e) ++j; return j; }
CLR_HD constexpr int tri_row(int e) { return e - tri_col(e) * (tri_col(e) + 1) / 2; }

// workgroup barrier that waits for nothing of this wave's own (T keeps its series prefetch in
// flight across it); "memory": the compiler may not move LDS accesses across it
__device__ __forceinline__ void split_barrier_() { asm volatile("s_barrier" ::: "memory"); }
#define split_barrier() do { if (!(dbg & 4)) split_barrier_(); } while (0)

__device__ __forceinline__ int hw_simd_id() {
  // HW_REG_HW_ID (id 4), SIMD_ID = bits [5:4]
  return __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
}

// ---- role T: the zero-start trajectory (C, b), the zero-start sums, and the published step ----
template <int JR, int JC, bool FAST>
__device__ __forceinline__ void split_trajectory(const Problem<JR, JC>& p, DirectSeries& src, int L, int n0,
                                                 int N, bool store, double* slot0 /* + lane */,
                                                 double* elem_out, double* ld0_out, double* q0_out,
                                                 int* flag0_out, int dbg) {
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  using Lk = SplitLink<JR, JC>;
  constexpr int SLOT_STRIDE = Lk::NPAY * 64;
  double b[J], C[SZ];
#pragma unroll
  for (int i = 0; i < J; ++i) b[i] = 0.0;
#pragma unroll
  for (int i = 0; i < SZ; ++i) C[i] = 0.0;

  double q0 = 0.0;
  LogProduct lp0;
  lp0.init();
  int flag0 = 0;
  // register pipeline of the series, PF steps deep (one coalesced 512-B wave load per array and
  // step; an HBM round trip is about one step long, so one step of prefetch is not enough)
  constexpr int PF = 4;
  double tq[PF + 1], dq[PF], yq[PF];  // tq[k] = t(i + 1 + k), dq[k] = diag(i + k), yq[k] = y(i + k)
  double tn = src.t(0);
#pragma unroll
  for (int k = 0; k < PF; ++k) { tq[k] = src.t(1 + k); dq[k] = src.diag(k); yq[k] = src.y(k); }
  for (int i = 0; i < L; ++i) {
    const double t_cur_next = tq[0], diag_cur = dq[0], y_cur = yq[0];
#pragma unroll
    for (int k = 0; k + 1 < PF; ++k) { tq[k] = tq[k + 1]; dq[k] = dq[k + 1]; yq[k] = yq[k + 1]; }
    if (!(dbg & 2)) {  // (reads past the chunk run into the next chunk / padding: masked by nleft)
      tq[PF - 1] = src.t(i + PF + 1);
      dq[PF - 1] = src.diag(i + PF);
      yq[PF - 1] = src.y(i + PF);
    }
    double* slot = slot0 + (i & 1) * SLOT_STRIDE;
    double u[J], v[J], phid[nz(JR + JC)];
    features_uv<JR, JC, FAST>(p, tn, u, v);
    features_phi_distinct<JR, JC>(p, t_cur_next - tn, phid);
    // publish as produced (no copies kept alive)
#pragma unroll
    for (int k = JR; k < J; ++k) slot[(Lk::F_U + k - JR) * 64] = u[k];
#pragma unroll
    for (int k = 0; k < JR + JC; ++k) slot[(Lk::F_PHI + k) * 64] = phid[k];

    double q[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) acc += C[sym(k, j)] * u[k];
      q[j] = acc;
    }
    double s = 0.0, ub = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) { s += u[j] * q[j]; ub += u[j] * b[j]; }
    const double D = p.diagonal(diag_cur) - s;
    const double invD = 1.0 / D;
    const double x = y_cur - ub;
    const bool valid = n0 + i < N;
    if (valid) {
      if (n0 + i >= 1 && !(D > 0.0)) flag0 = 1;
      lp0.mul(D);
      q0 += x * x * invD;
    }
    const double xs = x * invD;
    double z[J], W[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      z[j] = v[j] - q[j];
      W[j] = z[j] * invD;
      slot[(Lk::F_PW + j) * 64] = phid[phi_index<JR>(j)] * W[j];
    }
    slot[Lk::F_XS * 64] = xs;
    slot[Lk::F_INVD * 64] = invD;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    split_barrier();  // B(i): slot i & 1 is complete (and the rider has copied slot (i - 1) & 1 out)
#pragma unroll
    for (int j = 0; j < J; ++j) b[j] = phid[phi_index<JR>(j)] * (b[j] + W[j] * x);
    decay_rank1_update<JR, JC>(phid, z, W, C);
    tn = t_cur_next;
  }
  if (!store) return;
  *ld0_out = lp0.log_value();
  *q0_out = q0;
  *flag0_out = flag0;
  double* o = elem_out + J * J;
#pragma unroll
  for (int i = 0; i < J; ++i) o[i] = b[i];
  o += J;
#pragma unroll
  for (int i = 0; i < SZ; ++i) o[i] = C[i];
}

// ---- role R: A, eta (registers) and Jm (LDS) -----------------------------------------------------
template <int JR, int JC>
__device__ __forceinline__ void split_riders(const Problem<JR, JC>& p, int L, int n0, int N, bool store,
                                             const double* slot0 /* + lane */, double2* jm /* + lane */,
                                             double* elem_out, int dbg) {
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  using Lk = SplitLink<JR, JC>;
  double Acol[J * J], eta[J];  // Acol[j * J + i] = A[i][j]
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int i = 0; i < J; ++i) Acol[j * J + i] = (i == j) ? 1.0 : 0.0;
    eta[j] = 0.0;
  }
#pragma unroll
  for (int f = 0; f < Lk::NJM; ++f) jm[f * 64] = make_double2(0.0, 0.0);

  for (int i = 0; i < L; ++i) {
    split_barrier();  // B(i): step i is in slot i & 1
    const double* slot = slot0 + (i & 1) * (Lk::NPAY * 64);
    double u[J], pw[J], phid[nz(JR + JC)], r[J];
#pragma unroll
    for (int k = 0; k < JR; ++k) u[k] = p.ar[k];
#pragma unroll
    for (int k = JR; k < J; ++k) u[k] = slot[(Lk::F_U + k - JR) * 64];
#pragma unroll
    for (int k = 0; k < JR + JC; ++k) phid[k] = slot[(Lk::F_PHI + k) * 64];
#pragma unroll
    for (int k = 0; k < J; ++k) pw[k] = slot[(Lk::F_PW + k) * 64];
    const double invD = slot[Lk::F_INVD * 64];
    const double xs = slot[Lk::F_XS * 64];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool valid = n0 + i < N;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double racc = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) racc += Acol[j * J + k] * u[k];
      r[j] = racc;
#pragma unroll
      for (int k = 0; k < J; ++k)
        Acol[j * J + k] = phid[phi_index<JR>(k)] * Acol[j * J + k] - pw[k] * racc;
    }
    if (valid) {  // (padding steps of the short last chunk must not touch the accumulators)
      double rs[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        rs[j] = r[j] * invD;
        eta[j] -= r[j] * xs;
      }
#pragma unroll
      for (int f = 0; f < Lk::NJM; ++f) {
        double2 d = jm[f * 64];
        d.x -= r[tri_row(2 * f)] * rs[tri_col(2 * f)];
        if (2 * f + 1 < SZ) d.y -= r[tri_row(2 * f + 1)] * rs[tri_col(2 * f + 1)];
        jm[f * 64] = d;
      }
    }
  }
  if (!store) return;
  double* o = elem_out;
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = 0; j < J; ++j) o[i * J + j] = Acol[j * J + i];
  }
  o += J * J + J + SZ;
#pragma unroll
  for (int i = 0; i < J; ++i) o[i] = eta[i];
  o += J;
#pragma unroll
  for (int f = 0; f < Lk::NJM; ++f) {
    const double2 d = jm[f * 64];
    o[2 * f] = d.x;
    if (2 * f + 1 < SZ) o[2 * f + 1] = d.y;
  }
}

// 8 waves: 4 sets (problem b, block x of 64 chunks), one T and one R wave per set.
template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
summarize_split_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  using Lk = SplitLink<JR, JC>;
  __shared__ double ring[4][2 * Lk::NPAY * 64];  // two slots per set: T fills one while R reads the other
  __shared__ double2 jmbuf[4][Lk::NJM * 64];
  int* placed = reinterpret_cast<int*>(&jmbuf[0][0]);  // (LDS is full: borrowed until the roles are fixed)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 4) placed[threadIdx.x] = 0;
  __syncthreads();
  // role from the SIMD this wave actually runs on: first arrival = trajectory, second = riders
  const int simd = hw_simd_id();
  int slot_id = 0;
  if (lane == 0) slot_id = atomicAdd(&placed[simd], 1);
  slot_id = __builtin_amdgcn_readfirstlane(slot_id);
  __syncthreads();
  const int balanced = __builtin_amdgcn_readfirstlane(
      (placed[0] == 2 && placed[1] == 2 && placed[2] == 2 && placed[3] == 2) ? 1 : 0);
  // (everything below is wave-uniform: keep it in SGPRs so that the hyper-parameters are scalar loads)
  const int set = __builtin_amdgcn_readfirstlane(balanced ? simd : (wave & 3));
  const int role = __builtin_amdgcn_readfirstlane(balanced ? slot_id : (wave >> 2));  // 0 = T, 1 = R
  __syncthreads();  // (everybody has read `placed`: the riders may now clear their Jm cells)

  const int nblk = (P.nchunk + 63) / 64;
  const long g = (long)blockIdx.x * 4 + set;
  const bool live = g < (long)P.B * nblk;
  const int b = live ? (int)(g / nblk) : 0;
  const int xblk = live ? (int)(g % nblk) : 0;
  const int c = xblk * 64 + lane;
  const bool store = live && c < P.nchunk;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  const long cell = (long)b * P.nchunk + (store ? c : 0);
  double* elem = P.elems + cell * Wd::ELEM;
  double* slot = ring[set] + lane;
  if (role == 0) {
    DirectSeries src = make_direct(P, b, store ? c : 0);
    if (!store) src.nleft = 0;  // lanes past the last chunk / dead sets: padding only
    double ld0 = 0.0, q0 = 0.0;
    int flag0 = 0;
    split_trajectory<JR, JC, FAST>(p, src, P.L, c * P.L, P.N, store, slot, elem, &ld0, &q0, &flag0, P.split);
    if (store) {
      if (P.cond) { P.cond[cell * 3 + 0] = 0.0; P.cond[cell * 3 + 1] = 1.0; P.cond[cell * 3 + 2] = 0.0; }
      P.part[cell * 2 + 0] = ld0;
      P.part[cell * 2 + 1] = q0;
      P.flags[cell] = flag0;
    }
  } else {
    split_riders<JR, JC>(p, P.L, c * P.L, P.N, store, slot, jmbuf[set] + lane, elem, P.split);
  }
}

// host side: launch for shape (R, C) if it matches
template <int JR, int JC>
inline void launch_split_shape(const BatchParams& P, hipStream_t s) {
  const int nblk = (P.nchunk + 63) / 64;
  const long sets = (long)P.B * nblk;
  const dim3 grid((unsigned)((sets + 3) / 4)), block(512);
  if (P.fast_trig) hipLaunchKernelGGL((summarize_split_kernel<JR, JC, true>), grid, block, 0, s, P);
  else hipLaunchKernelGGL((summarize_split_kernel<JR, JC, false>), grid, block, 0, s, P);
}
#define CLR_SPLIT_SHAPE(R, C) if (JR == R && JC == C) { launch_split_shape<R, C>(P, s); return true; }

}  // namespace clr
