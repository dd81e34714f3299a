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
// relayout_kernel in api_kernels.hip, made once per set_series) instead of staging tiles.
//
// Synchronisation: R runs one step behind T, through TWO slots and one workgroup barrier per
// step: T fills slot i & 1 during step i and then meets R at barrier B(i); R, after B(i), reads
// slot i & 1 into registers and folds the step in while T computes step i + 1 into the other
// slot.  T may overwrite slot i & 1 again in step i + 2, i.e. after B(i + 1), which R only
// reaches once its reads of step i are done.  (Measured alternatives: two barriers around one
// slot 3.40 ms; a barrier-free hand-over through a marker in the slot, R polling with s_sleep,
// 3.31 ms; this 3.20 ms; without any synchronisation -- wrong results -- 2.94 ms.)
// The two waves of a SIMD share ONE issue port (fp64: 2.46 ns per instruction per SIMD at any wave count), so the
// step costs about the SUM of both instruction streams; what arbitration there is goes to the RIDERS (s_setprio,
// see the kernel): their step opens with LDS round trips, and unprioritised they lose the port to T's long fp64
// runs although T can never be more than one slot ahead (-8 %; profiles/r02b_split_notes.txt has the breakdown).
#pragma once

#include <type_traits>

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
#define split_barrier() split_barrier_()

__device__ __forceinline__ int hw_simd_id() {
  // HW_REG_HW_ID (id 4), SIMD_ID = bits [5:4]
  return __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
}

// ---- role T: the zero-start trajectory (C, b), the zero-start sums, and the published step ----
template <int JR, int JC, bool FAST>
__device__ __forceinline__ void split_trajectory(const Problem<JR, JC>& p, DirectSeries& src, int L, int n0,
                                                 int N, bool store, double* slot0 /* + lane */,
                                                 double* elem_out, double* ld0_out, double* q0_out,
                                                 int* flag0_out, double* gamma_out, int dbg) {
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  using Lk = SplitLink<JR, JC>;
  constexpr int SLOT_STRIDE = Lk::NPAY * 64;
  double gamma = 0.0;
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
  // The chunk-interleaved copy is padded past the end of the series (relayout_kernel: t held, diagonal 1e300,
  // y = 0), so the loads are unguarded and a padded step changes nothing above the rounding -- here or in the
  // rider wave (see split_trajectory_lazy for the argument); running wave-uniform offsets replace
  // DirectSeries::off per load.
  double tn = src.tp[src.off(0)];
#pragma unroll
  for (int k = 0; k < PF; ++k) { tq[k] = src.tp[src.off(1 + k)]; dq[k] = src.dp[src.off(k)]; yq[k] = src.yp[src.off(k)]; }
  long od = src.off(PF), ot = src.off(PF + 1);
  int id = PF;  // the index od stands for
  for (int i = 0; i < L; ++i) {
    const double t_cur_next = tq[0], diag_cur = dq[0], y_cur = yq[0];
#pragma unroll
    for (int k = 0; k + 1 < PF; ++k) { tq[k] = tq[k + 1]; dq[k] = dq[k + 1]; yq[k] = yq[k + 1]; }
    tq[PF - 1] = src.tp[ot];
    dq[PF - 1] = src.dp[od];
    yq[PF - 1] = src.yp[od];
    ++id;
    od = (id == L) ? src.cs : od + src.is;
    ot = (id + 1 == L) ? src.cs : ot + src.is;
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
    const double invD = recip_fast(D);
    const double x = y_cur - ub;
    {  // (a padded step has D ~ 1e300 > 0, a_n / D = 1 and x^2 / D ~ 0: only the log-determinant leaves it out)
      const int n = n0 + i;
      flag0 |= (n >= 1 && !(D > 0.0)) ? 1 : 0;
      lp0.mul_window(n < N ? D : 1.0);
      if ((i & 31) == 31) lp0.renorm();  // (i is wave-uniform: a scalar branch)
      q0 += x * x * invD;
      gamma = fmax(gamma, fabs(p.diagonal(diag_cur) * invD));
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
  *gamma_out = gamma;
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
    {  // (steps past the end of the series arrive with 1 / D ~ 1e-300 and x / D ~ 0 from the trajectory wave)
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

// ---------------------------------------------------------------------------------------------------
// LAZY decay (densely sampled series: max c * max dx < 2^-7, checked on the host).  The reference
// multiplies S, f (and here A) by the decays of every step (cholesky.h:154-160: 3 flops per entry).
// Factor the decay accumulated since the last renormalisation out of the state instead:
//     C = Psi Cbar Psi ,  b = Psi bbar ,  A = Psi Abar ,      Psi = diag(prod phi)
// Then with ubar = Psi u, vbar = Psi^-1 v:
//     q = Psi qbar, qbar = Cbar ubar ;  D = a - ubar.qbar ;  zbar = vbar - qbar ;  Wbar = zbar / D
//     x = y - ubar.bbar ;  r = Abar^T ubar ;  Abar -= Wbar r^T ;  Cbar += zbar Wbar^T ;  bbar += Wbar x
// and Psi <- Phi Psi: one FMA per state entry instead of FMA + MUL, the rider wave no longer needs phi
// or Phi W (18 published doubles instead of 21), ~110 fp64 instructions fewer per step of ~590.  Every
// RENORM steps (and at the end of the chunk) the state is multiplied out and Psi reset to 1: 64 since round 5 (16
// before: headline summarize 2.169 -> 2.14 ms, the log determinants' checksum unchanged to 13 digits), so Psi stays
// within [0.6, 1] and Psi^-1 (accumulated beside it from exp(+c dx)) within [1, 1.65]: the scaling is
// rounding-neutral, and the drift of Psi * Psi^-1 from 1 is bounded by 64 roundings.
// ---------------------------------------------------------------------------------------------------
template <int JR, int JC>
struct SplitLinkLazy {
  static constexpr int J = JR + 2 * JC;
  static constexpr int F_U = 0, F_W = J, F_INVD = 2 * J, F_Y = 2 * J + 1, NPAY = 2 * J + 2;
  static constexpr int M = JR + JC;  // distinct decays: one slot area of M doubles per lane for Psi
#ifndef CLR_SPLIT_RENORM
#define CLR_SPLIT_RENORM 64
#endif
  static constexpr int RENORM = CLR_SPLIT_RENORM;
};

// phi = exp(-c dx) and 1/phi = exp(+c dx) of the distinct decays, sharing the even / odd parts
template <int JR, int JC>
__device__ __forceinline__ void features_phi_pair(const Problem<JR, JC>& p, double dx, double* phid, double* phinv) {
  constexpr int M = JR + JC;
  double x[nz(M)];
  double amax = 0.0;
#pragma unroll
  for (int j = 0; j < JR; ++j) { x[j] = -p.cr[j] * dx; amax = fmax(amax, fabs(x[j])); }
#pragma unroll
  for (int j = 0; j < JC; ++j) { x[JR + j] = -p.cc[j] * dx; amax = fmax(amax, fabs(x[JR + j])); }
  if (CLR_WAVE_ALL(amax < 0.0009765625)) {  // 2^-10: even part to x^4, odd part to x^3 (next terms < 7.4e-18)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double x2 = x[j] * x[j];
      const double ch = fma(x2, fma(x2, 1.0 / 24.0, 0.5), 1.0);
      const double sh = x[j] * fma(x2, 1.0 / 6.0, 1.0);
      phid[j] = ch + sh;
      phinv[j] = ch - sh;
    }
  } else if (CLR_WAVE_ALL(amax < 0.0078125)) {  // 2^-7: to x^6 / x^5 (next term x^7/5040 < 3.5e-19)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double x2 = x[j] * x[j];
      const double ch = fma(x2, fma(x2, fma(x2, 1.0 / 720.0, 1.0 / 24.0), 0.5), 1.0);
      const double sh = x[j] * fma(x2, fma(x2, 1.0 / 120.0, 1.0 / 6.0), 1.0);
      phid[j] = ch + sh;
      phinv[j] = ch - sh;
    }
  } else {  // (the host only selects the lazy kernels for dense series; kept for safety)
#pragma unroll
    for (int j = 0; j < M; ++j) { phid[j] = exp(x[j]); phinv[j] = exp(-x[j]); }
  }
}

template <int JR, int JC, bool FAST>
__device__ __forceinline__ void split_trajectory_lazy(const Problem<JR, JC>& p, DirectSeries& src, int L, int n0,
                                                      int N, bool store, double* slot0 /* + lane */,
                                                      double* psi_area /* + lane */, double* elem_out,
                                                      double* ld0_out, int* flag0_out, double* gamma_out, int dbg) {
  // (in the lazy variant the zero-start f-trajectory bbar, x and the quadratic sum live in the RIDER
  //  wave: they need only ubar, Wbar, 1/D and y, and the registers they free here pay for Psi, Psi^-1)
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  using Lk = SplitLinkLazy<JR, JC>;
  constexpr int M = Lk::M;
  constexpr int SLOT_STRIDE = Lk::NPAY * 64;
  double C[SZ], psi[nz(M)], psinv[nz(M)];
#pragma unroll
  for (int i = 0; i < SZ; ++i) C[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) { psi[i] = 1.0; psinv[i] = 1.0; }
  double gamma = 0.0;
  LogProduct lp0;
  lp0.init();
  int flag0 = 0;
  constexpr int PF = 3;
  double tq[PF + 1], dq[PF], yq[PF];
  // Steps past the end of the series (the tail of a problem's last chunk) are HARMLESS instead of masked: the
  // chunk-interleaved copy this kernel reads is padded by relayout_kernel with t = the last sample (dx = 0: no
  // decay, no rotation), diagonal = 1e300 (1 / D ~ 1e-300) and y = 0, so every update of such a step -- here and
  // in the rider wave -- is below the rounding of what it is added to, and neither wave needs selects or guarded
  // loads (the sums of this wave are still restricted to n < N below; reads past the chunk's end land in the next
  // chunk or, for the last chunk, in rows 1..5 of the copy: only the state after the last step sees them, and the
  // last chunk's end state is not used).  Dead lanes (no chunk of their own) read chunk 0's samples.
  auto tt = [&](int i) { return src.tp[src.off(i)]; };
  auto dd = [&](int i) { return src.dp[src.off(i)]; };
  auto yy = [&](int i) { return src.yp[src.off(i)]; };
  double tn = tt(0);
  double cdv[nz(JC)], sdv[nz(JC)];  // cos / sin of d t at the current sample
#pragma unroll
  for (int j = 0; j < JC; ++j) { cdv[j] = 1.0; sdv[j] = 0.0; }
#pragma unroll
  for (int k = 0; k < PF; ++k) { tq[k] = tt(1 + k); dq[k] = dd(k); yq[k] = yy(k); }
  // running (wave-uniform) offsets of the samples fetched next: index i + PF for the diagonal and y, one more for t;
  // they advance by the sample stride and jump into the next chunk at index L (DirectSeries::off without the
  // per-step 64-bit multiplies)
  long od = src.off(PF), ot = src.off(PF + 1);
  int id = PF;  // the index od stands for
  for (int i0 = 0; i0 < L; i0 += Lk::RENORM) {
    const int i1 = (i0 + Lk::RENORM < L) ? i0 + Lk::RENORM : L;
    // anchor: the full sincos of the absolute phase at the block's first sample (cholesky.h:137); at
    // most RENORM - 1 = 63 rotations (~1e-14 absolute; 15 until round 5) accumulate before the next anchor.
    // (The series queue's shift costs 9 v_mov_b64 per step.  Three steps per trip -- the rotation closes after PF
    //  steps and the copies vanish -- was tried in round 5: the compiler then hoists the next step's
    //  state-independent work over the barrier, 58 registers spill, 2.20 -> 3.85 ms.)
#pragma unroll
    for (int j = 0; j < JC; ++j) sincos_phase<FAST>(p.dc[j] * tn, &sdv[j], &cdv[j]);
    // one trajectory step on the sample (t_cur_next, diag_cur, y_cur) handed in (the caller owns the series queue)
    auto one_step = [&](int i, const double t_cur_next, const double diag_cur, const double y_cur) __attribute__((always_inline)) {
      double* slot = slot0 + (i & 1) * SLOT_STRIDE;
      slot[Lk::F_Y * 64] = y_cur;
      double u[J], v[J];
      // U~, V~ (cholesky.h:129-147) from the block's running (cos, sin) pairs
#pragma unroll
      for (int j = 0; j < JR; ++j) { u[j] = p.ar[j]; v[j] = 1.0; }
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        const int k = JR + 2 * j;
        u[k] = p.ac[j] * cdv[j] + p.bc[j] * sdv[j];
        u[k + 1] = p.ac[j] * sdv[j] - p.bc[j] * cdv[j];
        v[k] = cdv[j];
        v[k + 1] = sdv[j];
      }
      // ... which advance to the next sample by a ROTATION through the small angle d dx (12 fp64
      // instructions per term instead of 21 + 14 integer ones: no range reduction, no quadrant
      // selects); |d dx| < 2^-5 is checked on the host for the whole plan (lazy_eligible)
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        const double dl = p.dc[j] * (t_cur_next - tn), d2 = dl * dl;
        const double sn = dl * fma(d2, fma(d2, fma(d2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
        const double cs = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
        const double c0 = cdv[j], s0 = sdv[j];
        cdv[j] = fma(c0, cs, -s0 * sn);
        sdv[j] = fma(s0, cs, c0 * sn);
      }
#pragma unroll
      for (int k = 0; k < J; ++k) {
        u[k] *= psi[phi_index<JR>(k)];                                               // ubar
        v[k] = (k < JR) ? psinv[phi_index<JR>(k)] : v[k] * psinv[phi_index<JR>(k)];  // vbar (v = 1 on real rows)
        slot[(Lk::F_U + k) * 64] = u[k];
      }
      {  // Psi for the NEXT step (this step's decay included); the old one is no longer needed
        double phid[nz(M)], phinv[nz(M)];
        features_phi_pair<JR, JC>(p, t_cur_next - tn, phid, phinv);
#pragma unroll
        for (int m = 0; m < M; ++m) { psi[m] *= phid[m]; psinv[m] *= phinv[m]; }
      }
      double q[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) acc += C[sym(k, j)] * u[k];
        q[j] = acc;
      }
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) s += u[j] * q[j];
      const double a_n = p.diagonal(diag_cur);
      const double D = a_n - s;
      const double invD = recip_fast(D);
      {  // (a padded step has D ~ 1e300 > 0 and a_n / D = 1: only the log-determinant has to leave it out)
        const int n = n0 + i;
        flag0 |= (n >= 1 && !(D > 0.0)) ? 1 : 0;
        lp0.mul_window(n < N ? D : 1.0);
        gamma = fmax(gamma, fabs(a_n * invD));
      }
      double z[J], W[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        z[j] = v[j] - q[j];   // zbar
        W[j] = z[j] * invD;   // Wbar
        slot[(Lk::F_W + j) * 64] = W[j];
      }
      slot[Lk::F_INVD * 64] = invD;
      if (i + 1 == i1) {  // last step of the block: the rider multiplies Psi out after folding it in
#pragma unroll
        for (int m = 0; m < M; ++m) psi_area[m * 64] = psi[m];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      split_barrier();  // B(i)
#ifdef CLR_SPLIT_UNROLL3
      // (three steps per loop trip: nothing of the NEXT step may be computed on this side of the barrier -- round 5's
      //  attempt died of exactly that hoisting, 58 registers spilled -- so the queue's registers pass through an opaque
      //  asm here and the scheduler is fenced)
      asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(yq[0]), "+v"(yq[1]), "+v"(yq[2]));
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) C[tri(k, j)] = fma(z[k], W[j], C[tri(k, j)]);
      }
      tn = t_cur_next;
    };
    auto refill = [&](int k) __attribute__((always_inline)) {  // entry k of the queue <- the sample PF steps ahead
      tq[k] = src.tp[ot];
      dq[k] = src.dp[od];
      yq[k] = src.yp[od];
      ++id;
      od = (id == L) ? src.cs : od + src.is;
      ot = (id + 1 == L) ? src.cs : ot + src.is;
    };
    int i = i0;
#ifdef CLR_SPLIT_UNROLL3
    static_assert(PF == 3, "the unrolled trip rotates a queue of three samples");
    // The queue by INDEX instead of by shifting (9 v_mov_b64 per step): step i uses entry i mod 3 and refills it with
    // sample i + 3; after three steps the rotation has closed.  (RENORM is not a multiple of 3: the block's last one
    // or two steps take the shifting form below, on a queue that is in canonical order again.)
    for (; i + 3 <= i1; i += 3) {
      { const double a = tq[0], b_ = dq[0], c_ = yq[0]; refill(0); one_step(i, a, b_, c_); }
      { const double a = tq[1], b_ = dq[1], c_ = yq[1]; refill(1); one_step(i + 1, a, b_, c_); }
      { const double a = tq[2], b_ = dq[2], c_ = yq[2]; refill(2); one_step(i + 2, a, b_, c_); }
    }
#endif
    for (; i < i1; ++i) {
      const double t_cur_next = tq[0], diag_cur = dq[0], y_cur = yq[0];
#pragma unroll
      for (int k = 0; k + 1 < PF; ++k) { tq[k] = tq[k + 1]; dq[k] = dq[k + 1]; yq[k] = yq[k + 1]; }
      refill(PF - 1);
      one_step(i, t_cur_next, diag_cur, y_cur);
    }
    lp0.renorm();  // (RENORM factors in [0.5, 1) since the last one)
    {  // multiply the accumulated decay out (the rider does the same to Abar, bbar)
      double pp[nz(M * (M + 1) / 2)];
#pragma unroll
      for (int bb = 0; bb < M; ++bb) {
#pragma unroll
        for (int aa = 0; aa <= bb; ++aa) pp[tri(aa, bb)] = psi[aa] * psi[bb];
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) C[tri(k, j)] *= pp[tri(phi_index<JR>(k), phi_index<JR>(j))];
      }
#pragma unroll
      for (int m = 0; m < M; ++m) { psi[m] = 1.0; psinv[m] = 1.0; }
    }
  }
  if (!store) return;
  *ld0_out = lp0.log_value();
  *flag0_out = flag0;
  *gamma_out = gamma;
  double* o = elem_out + J * J + J;
#pragma unroll
  for (int i = 0; i < SZ; ++i) o[i] = C[i];
}

template <int JR, int JC>
__device__ __forceinline__ void split_riders_lazy(int L, int n0, int N, bool store, const double* slot0 /* + lane */,
                                                  const double* psi_area /* + lane */, double2* jm /* + lane */,
                                                  double* elem_out, double* q0_out, int dbg) {
  constexpr int J = Widths<JR, JC>::J;
  constexpr int SZ = Widths<JR, JC>::SZ;
  using Lk = SplitLinkLazy<JR, JC>;
  using LkJ = SplitLink<JR, JC>;
  constexpr int M = Lk::M;
  double Acol[J * J], eta[J], b[J];  // Acol[j * J + i] = Abar[i][j]; b = bbar
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int i = 0; i < J; ++i) Acol[j * J + i] = (i == j) ? 1.0 : 0.0;
    eta[j] = 0.0;
    b[j] = 0.0;
  }
  double q0 = 0.0;
#pragma unroll
  for (int f = 0; f < LkJ::NJM; ++f) jm[f * 64] = make_double2(0.0, 0.0);
  // Jm -= r r^T / D is a pure accumulation (never read inside the loop) and its read-modify-write is most of the
  // riders' LDS traffic (37 of 45 KB per wave and step).  The steps go in PAIRS: the first keeps its (r, 1/D) in
  // the 18 registers the riders have left, the second folds both rank-one terms in with one pass over the cells.
  // Steps past the end of the series contribute exact zeros (selects, no branch).  Worth 1.5 % once the riders have
  // the issue priority (2.53 -> 2.49 ms, same box; nothing before that: profiles/r02b_split_notes.txt).
  double rp[J], invDp = 0.0;
  // MODE 0: first of a pair (defer); 1: second of a pair (fold both); 2: a lone last step of the block
  auto step = [&](int i, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    split_barrier();  // B(i)
    const double* slot = slot0 + (i & 1) * (Lk::NPAY * 64);
    double u[J], W[J], r[J];
#pragma unroll
    for (int k = 0; k < J; ++k) u[k] = slot[(Lk::F_U + k) * 64];
#pragma unroll
    for (int k = 0; k < J; ++k) W[k] = slot[(Lk::F_W + k) * 64];
    const double invD = slot[Lk::F_INVD * 64];
    const double y = slot[Lk::F_Y * 64];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double ub = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) ub += u[k] * b[k];
    const double x = y - ub;           // cholesky.h:353-355 from the zero start
    const double xs = x * invD;
#pragma unroll
    for (int k = 0; k < J; ++k) b[k] = fma(W[k], x, b[k]);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double racc = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) racc += Acol[j * J + k] * u[k];
      r[j] = racc;
#pragma unroll
      for (int k = 0; k < J; ++k) Acol[j * J + k] = fma(-W[k], racc, Acol[j * J + k]);
    }
    // (steps past the end of the series arrive with 1/D ~ 1e-300 from the trajectory wave: no masking needed)
    const double vinvD = invD, vxs = xs;
    q0 += x * xs;
#pragma unroll
    for (int j = 0; j < J; ++j) eta[j] = fma(-r[j], vxs, eta[j]);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < J; ++j) rp[j] = r[j];
      invDp = vinvD;
    } else {
      double rs[J], rsp[J];
#pragma unroll
      for (int j = 0; j < J; ++j) { rs[j] = r[j] * vinvD; rsp[j] = (MODE == 1) ? rp[j] * invDp : 0.0; }
#pragma unroll
      for (int f = 0; f < LkJ::NJM; ++f) {
        double2 d = jm[f * 64];
        if (MODE == 1) d.x = fma(-rp[tri_row(2 * f)], rsp[tri_col(2 * f)], d.x);
        d.x = fma(-r[tri_row(2 * f)], rs[tri_col(2 * f)], d.x);
        if (2 * f + 1 < SZ) {
          if (MODE == 1) d.y = fma(-rp[tri_row(2 * f + 1)], rsp[tri_col(2 * f + 1)], d.y);
          d.y = fma(-r[tri_row(2 * f + 1)], rs[tri_col(2 * f + 1)], d.y);
        }
        jm[f * 64] = d;
        if ((f % 6) == 5) __builtin_amdgcn_sched_barrier(0);  // (keep the 18 cells from being loaded all at once)
      }
    }
  };
  for (int i0 = 0; i0 < L; i0 += Lk::RENORM) {
    const int i1 = (i0 + Lk::RENORM < L) ? i0 + Lk::RENORM : L;
    int i = i0;
    for (; i + 1 < i1; i += 2) {
      step(i, std::integral_constant<int, 0>());
      step(i + 1, std::integral_constant<int, 1>());
    }
    if (i < i1) step(i, std::integral_constant<int, 2>());
    {  // the block's accumulated decay (published by T with the block's last step; T rewrites it 16 steps on)
      double psi[nz(M)];
#pragma unroll
      for (int m = 0; m < M; ++m) psi[m] = psi_area[m * 64];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        b[j] *= psi[phi_index<JR>(j)];
#pragma unroll
        for (int k = 0; k < J; ++k) Acol[j * J + k] *= psi[phi_index<JR>(k)];
      }
    }
  }
  if (!store) return;
  *q0_out = q0;
  double* o = elem_out;
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = 0; j < J; ++j) o[i * J + j] = Acol[j * J + i];
  }
  o += J * J;
#pragma unroll
  for (int i = 0; i < J; ++i) o[i] = b[i];
  o += J + SZ;
#pragma unroll
  for (int i = 0; i < J; ++i) o[i] = eta[i];
  o += J;
#pragma unroll
  for (int f = 0; f < LkJ::NJM; ++f) {
    const double2 d = jm[f * 64];
    o[2 * f] = d.x;
    if (2 * f + 1 < SZ) o[2 * f + 1] = d.y;
  }
}

// SETS == 4: 8 waves = 4 sets (problem b, block x of 64 chunks), one T and one R wave per set, the per-step barrier
// spans all 8 waves.  SETS == 1 (round 4 A/B, profiles/r04g_split_wg_ab.txt): one set per 128-thread workgroup -- the
// barrier couples only the T and R wave that actually exchange data; four workgroups per CU.
template <int JR, int JC, bool FAST, bool LAZY, int SETS = 4>
__global__ void __launch_bounds__(128 * SETS) __attribute__((amdgpu_waves_per_eu(2, 2)))
summarize_split_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  using Lk = SplitLink<JR, JC>;
  using LkL = SplitLinkLazy<JR, JC>;
  constexpr int NPAY = LAZY ? LkL::NPAY : Lk::NPAY;
  __shared__ double ring[SETS][2 * NPAY * 64];  // two slots per set: T fills one while R reads the other
  __shared__ double psibuf[LAZY ? SETS : 1][(LAZY ? LkL::M : 1) * 64];  // Psi of the renormalisation steps
  __shared__ double2 jmbuf[SETS][Lk::NJM * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int set = 0, role = wave;  // (SETS == 1: the first wave is the trajectory, the second the riders)
  if (SETS == 4) {
    int* placed = reinterpret_cast<int*>(&jmbuf[0][0]);  // (LDS is full: borrowed until the roles are fixed)
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
    set = __builtin_amdgcn_readfirstlane(balanced ? simd : (wave & 3));
    role = __builtin_amdgcn_readfirstlane(balanced ? slot_id : (wave >> 2));  // 0 = T, 1 = R
    __syncthreads();  // (everybody has read `placed`: the riders may now clear their Jm cells)
  }

  const int nblk = (P.nchunk + 63) / 64;
  const long g = (long)blockIdx.x * SETS + set;
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
  // The rider wave gets the issue priority: its step starts with LDS round trips (the published step, then Jm),
  // so it has the fewer instructions ready at any moment and loses the arbitration to the trajectory's long
  // fp64 runs -- yet the trajectory cannot run ahead of it by more than one slot.  Measured on one box, same
  // run: 3.08 ms default, 3.08 ms with T prioritised, 2.85 ms with R prioritised (profiles/r02b_split_notes.txt).
  if (role == 1) __builtin_amdgcn_s_setprio(3);
  if (role == 0) {
    DirectSeries src = make_direct(P, b, store ? c : 0);
    if (!store) src.nleft = 0;  // lanes past the last chunk / dead sets: padding only
    double ld0 = 0.0, q0 = 0.0, gamma = 0.0;
    int flag0 = 0;
    if (LAZY)
      split_trajectory_lazy<JR, JC, FAST>(p, src, P.L, c * P.L, P.N, store, slot, psibuf[LAZY ? set : 0] + lane, elem,
                                          &ld0, &flag0, &gamma, P.split);
    else
      split_trajectory<JR, JC, FAST>(p, src, P.L, c * P.L, P.N, store, slot, elem, &ld0, &q0, &flag0, &gamma, P.split);
    if (store) {
      if (P.cond) { P.cond[cell * 3 + 0] = gamma; P.cond[cell * 3 + 1] = 1.0; P.cond[cell * 3 + 2] = 0.0; }
      P.part[cell * 2 + 0] = ld0;
      if (!LAZY) P.part[cell * 2 + 1] = q0;  // (lazy: the rider wave owns the quadratic sum)
      P.flags[cell] = flag0;
    }
  } else {
    if (LAZY) {
      double q0 = 0.0;
      split_riders_lazy<JR, JC>(P.L, c * P.L, P.N, store, slot, psibuf[LAZY ? set : 0] + lane, jmbuf[set] + lane, elem,
                                &q0, P.split);
      if (store) P.part[cell * 2 + 1] = q0;
    } else
      split_riders<JR, JC>(p, P.L, c * P.L, P.N, store, slot, jmbuf[set] + lane, elem, P.split);
  }
}

// host side: launch for shape (R, C) if it matches
template <int JR, int JC>
inline void launch_split_shape(const BatchParams& P, hipStream_t s) {
  const int nblk = (P.nchunk + 63) / 64;
  const long sets = (long)P.B * nblk;
#ifndef CLR_SPLIT_SETS
#define CLR_SPLIT_SETS 4
#endif
  const dim3 grid((unsigned)((sets + CLR_SPLIT_SETS - 1) / CLR_SPLIT_SETS)), block(128 * CLR_SPLIT_SETS);
#define CLR_GO(F, Z) hipLaunchKernelGGL((summarize_split_kernel<JR, JC, F, Z, CLR_SPLIT_SETS>), grid, block, 0, s, P)
  if (P.split_lazy) { if (P.fast_trig) CLR_GO(true, true); else CLR_GO(false, true); }
  else              { if (P.fast_trig) CLR_GO(true, false); else CLR_GO(false, false); }
#undef CLR_GO
}
#define CLR_SPLIT_SHAPE(R, C) if (JR == R && JC == C) { launch_split_shape<R, C>(P, s); return true; }

}  // namespace clr
