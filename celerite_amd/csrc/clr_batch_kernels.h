// celerite_amd/csrc/clr_batch_kernels.h
//
// gfx950 kernels of the batched (problem x chunk) scan; see clr_core.h for the
// algebra.  Lane mapping:
//   summarize / replay : grid (ceil(nchunk/64), B), one 64-lane wave per block;
//       lane = chunk index within the problem blockIdx.y.  All 64 lanes of a wave
//       work on the SAME problem, so its hyper-parameters are wave-uniform (SGPRs,
//       scalar loads) and only the recurrence state lives in VGPRs.
//   prefix / finalize  : one lane per problem (sequential over that problem's
//       chunks; the work is O(nchunk J^3), a few percent of the total).
// HBM traffic: 24 B per sample per pass (t, diag, y).  A lane streaming its own
// 8 B/step run of the row-major arrays needs every 64-B line to survive 8 steps
// in L1/L2 (measured: replay 2x slower at 2 waves/SIMD), so the series are read
// either through StagedSeries (default: cooperative coalesced tiles transposed in
// LDS, no extra pass) or from a chunk-interleaved copy [problem][i][chunk] made by
// relayout_kernel (api_kernels.hip; one coalesced 512-B wave load per array per step).
// The workspace (elements, start states, partial sums) is O(nchunk) per problem.
#pragma once

#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>

#include "clr_core.h"

namespace clr {

constexpr int CLR_OK_STATUS = 0;  // (= CLR_OK of include/celerite_hip.h, which this header does not include)
constexpr int CLR_PENDING_STATUS = -1;  // internal: left to the scan pipeline by the warm path; never handed out
struct BatchParams {
  int B, N, nchunk, L;
  // general semiseparable terms (cholesky.h:65-72,114-116,148-152), wide path only: J_general extra rows behind the
  // celerite rows with phi = 1 and per-sample features U[j][n], V[j][n] (row-major [J_general][N] per problem), A[n]
  // added to the diagonal; strides in doubles between problems (0: shared)
  int J_general;
  const double *gen_A, *gen_U, *gen_V;
  long gen_A_stride, gen_U_stride, gen_V_stride;
  int L0;  // wide path: samples of the FIRST chunk when it differs from L (its summarize carries no riders and is
           // given more samples in return); 0: uniform chunks
  const double *jitter, *a_real, *c_real, *a_comp, *b_comp, *c_comp, *d_comp;
  // series as the kernels read them.  staged == 0 (DirectSeries): element (chunk c,
  // local i) of problem b is at base[b * stride + c * lane_cs + i * lane_is].
  // staged == 1 (StagedSeries): the row-major API arrays, base[b * stride + n].
  const double *t, *diag, *y;
  long t_stride, diag_stride, y_stride;
  long lane_is, lane_cs;
  int staged;
  int fast_trig;  // host-verified: max|d_comp| * max|t| < CLR_FAST_TRIG_LIMIT
  int split;      // summarize as two roles on two waves per SIMD (widths 7, 8; clr_split_kernels.h);
                  // needs the chunk-interleaved series (staged == 0, lane_cs == 1)
  double* scan_ws;  // wide layout: workspace of the parallel prefix (wide_prefix_scan.hip), null = the sequential walk
  int coop_prefix;  // prefix phase: 0 one lane per problem (the reference version), 1 16 lanes per problem walking
                    // the chunks in order, 2 multi-level (clr_prefix_kernels.h) following `plan`
  PrefixPlan plan;        // levels of the multi-level prefix (plan.levels == 0: the plain walk)
  double* lvl_elems;      // composed elements of levels 1..plan.levels, [B][plan.n[l]][ELEM] back to back
  double* lvl_starts;     // their start states, [B][plan.n[l]][START] back to back
  double* elems;   // [B][nchunk][ELEM]
  double* starts;  // [B][nchunk][START]
  // replay-free path: summarize writes each chunk's ZERO-START sums, correct_kernel
  // adds the chunk_update corrections and raises need_exact[b] for suspicious problems
  double* part;    // [B][nchunk][2]  (sum log D, sum x^2/D)
  double* cond;    // [B][nchunk][3]  conditioning record, may be null: max a_n / D0_n of the chunk's
                   //                 zero-start pivots (summarize), certificate pivot mu (correct), and
                   //                 the replay's end state against the scanned start of the next chunk
                   //                 (relative residual; replay_kernel)
  double* egerr;   // [B][nchunk]     measured accuracy of the chunk's G = (I + P Jm)^-1 P (correct kernels; chunk_update's
                   //                 eg_out: first-order forward error from the residual, relative), may be null
  int* flags;      // [B][nchunk]     zero-start pivot <= 0 seen / chunk suspicious
  int* need_exact; // [B]             0 settled from the chunk summaries; 1 ill-conditioned: chunked replay
                   //                 with its end states checked against the scanned starts; >= 2
                   //                 (certificate failed / that check failed): sequential recurrence
  // exact path (replay): only for problems with need_exact, or all if force_exact
  double* partx;   // [B][nchunk][2]
  int* flagsx;     // [B][nchunk]     D_n < 0 (n >= 1) seen: cholesky.h:176
  int force_exact; // materialising runs and clr_batch_set_exact(h, 1)
  int wide_materialize;  // wide path, MODE 0: also write phi, u, W, D of problem b in the reference's storage
                         // (cholesky.h:76-78, :703-706; CholeskySolver.compute at widths 9..64)
  int split_lazy;     // role-split summarize with the decay factored out of the state (dense series only)
  int seq_only;       // wide path: this launch only walks the problems with need_exact != 0 (one chunk = all N)
  // Materialising runs, refinement of the chunk heads (clr_batch_set_factor_refine): the materialising replay writes the
  // state it reaches at every chunk's END to `ends`; a second, short launch of the same kernel (`fixup_steps` > 0)
  // recomputes the first fixup_steps samples of every chunk c >= 1 from ends[c - 1] -- the reference recurrence carried
  // across the boundary -- instead of from the scanned start state, whose rounding (the scan algebra's, amplified by the
  // cancellation in D = a - u^T S u) otherwise shows in W and D for the ~32 samples the recurrence needs to forget it
  // (profiles/r06d_factor_error.txt: 4e-11 at a chunk's first sample, 4e-13 from sample 32 on -- the sequential
  // oracle's own distance from the binary128 truth).
  double* ends;       // [B][nchunk][START] or null
  int fixup_steps;    // > 0: this launch is the fix-up pass
  int dense;          // the plan's series are densely sampled: max |d| x max dx < 2^-5 (the phases may advance by rotations)
  int refine_samples; // the setting (samples per chunk head) the flow launches the fix-up pass with
  int defer_level1;   // problems the conditioning record sends to the checked chunked replay (level 1) are NOT replayed
                      // inline -- one flagged problem would cost the whole batch a sequential chunk-time -- but left with
                      // a pending status: the host re-plans them as a small plan of their own with many short chunks
                      // before results are handed out (api_batch.hip: rescue_run)
  int logdet_only;    // no right-hand side (CholeskySolver.compute): the quadratic form is not checked
  double cert_gamma;  // a problem whose conditioning record gamma_max / mu_min reaches this leaves the
                      // replay-free route (decide_kernel); <= 0: never
  double cert_gamma_abs;  // ... or whose gamma_max alone reaches this (<= 0: no such test)
  double cert_eg;         // ... or whose gamma_max x (largest measured G error of its chunks) reaches this (<= 0: no test)
  double cert_resid;  // largest relative mismatch between a replayed chunk's end state and the scanned
                      // start state of the next chunk that still counts as consistent
  // Output check (round 6).  A state mismatch above cert_resid need not show in any output: with IDENTICAL (or nearly
  // identical) terms -- the reference's own benchmark kernels, examples/benchmark/run.py:80-84 -- the state has directions
  // no u_n ever probes, rounding of the scan algebra and of the recurrence collect there without being forgotten, and the
  // END-STATE test sends a problem whose factor agrees with the oracle to 1e-12 to the sequential recurrence (width 16,
  // N = 65536: 29.6 ms instead of 0.6; profiles/r06q_family_factor.txt).  Such a problem (level 3: state mismatch in
  // (cert_resid, head_cap]) is replayed again, every chunk c >= 1 from the END state of chunk c - 1's previous replay --
  // the recurrence carried across the boundary -- and what the two replays WROTE is compared: D relatively, W against
  // the chunk's largest entry.  Within head_tol: settled (level 1) -- consecutive replays that agree are the sequential
  // recurrence's own factor to that tolerance (what the outputs see of a start state is forgotten along a chunk; replay
  // k is exact up to chunk k) -- and the last replay's entries, sums and flags stand; else once more from the new end
  // states (`ends_in` / `ends` swap), after the last attempt the sequential recurrence.
  int head_check;     // this fix-up launch is that check (2: the last attempt): level-3 problems only, cond[.][2] = the output mismatch
  const double* ends_in;  // ... the end states it starts from (`ends`: the ones it writes)
  double* ends_alt;       // the flow's second buffer of end states ([B][nchunk][START]; null: no output check)
  double head_cap;    // > 0: the route is on (check_replay kernels raise level 3 instead of 2 up to this state mismatch)
  double head_tol;    // largest output mismatch that counts as agreement
  // chunk-parallel gradient (clr_grad_kernels.h): riders [B][nchunk][RID], records [B][nchunk][NG][OUT], result [B][NG]
  double *g_riders, *g_out, *g_res;
  int g_m, g_nchunk;      // a gradient chunk = g_m chunks of the scan; g_nchunk = ceil(nchunk / g_m)
  // reverse mode: per-sample record w, D, x [B][step][J + 2][chunk], stored states every g_K steps
  // [B][checkpoint][SZ + J][chunk], state after / adjoint at the end of every chunk, per-chunk partials and drift
  double *g_rec, *g_ck, *g_ends, *g_adj, *g_adj0, *g_part, *g_drift, *g_drift_max;
  double* g_slab;                    // [B][slabs of 256 gradient chunks][33]: partial sums of the partials | the slab's certificate
  double *g_grp_riders, *g_grp_adj;  // two-level adjoint walk: composed riders / end adjoints of groups of g_seg gradient chunks
  int g_seg;                         // 0: one walk over all gradient chunks
  long g_rec_stride, g_ck_stride;  // doubles per problem
  int g_K;                // stored states: every g_K steps (> 0), or where the accumulated decay asks for one (0: GradStore)
  int g_nalloc;           // slots per chunk in g_ck
  int g_span;             // (adaptive rule) stored states at least this many steps apart, the rest rebuilt forwards (GradStore::span)
  unsigned char* g_ckflag;  // [B][waves of 64 chunks][steps]: what the forward pass did before each step (GradStore)
  double* g_count;        // [B][g_nchunk]: slots each chunk used
  int g_from_elems;       // reverse mode, g_m == 1: riders from the scan's elements (grad_riders_elem_kernel)
  const int* g_mask;      // forward-mode kernels: only the problems with g_mask[b] != 0 (null: all)
  // warm-started plain recurrence (warm_kernel; series that forget their past): its own chunking and workspace
  const int* wK;     // [B] warm-up steps of problem b (wave-uniform per block); <= 0: the problem takes the scan
  // the series as the warm kernel reads them: [problem][row][chunk], row r of chunk c = sample c wL - wKpad + r,
  // wrows = wKpad + wL + 8 rows; samples outside the series are padding a recurrence step ignores (relayout_warm_kernel)
  const double *wt, *wdiag, *wy;
  long wt_stride, wdiag_stride, wy_stride;  // per problem (0: one shared series)
  int wKpad, wrows;
  int wL, wnchunk;   // chunk length / count of the warm path
  double* wstarts;   // [B][wnchunk][START] state after the warm-up = at the chunk's first sample
  double* wends;     // [B][wnchunk][START] state after the chunk = at the next chunk's first sample
  double* wpart;     // [B][wnchunk][2]
  int* wflags;       // [B][wnchunk]
  double* wresid;    // [B] largest boundary mismatch found by warm_check_kernel
  int* need_scan;    // [B] warm_check_kernel: 1 = not settled by the warm path (the scan pipeline must run for it)
  int only_pending;  // the scan pipeline runs behind the warm path: only for problems with need_scan != 0
  double warm_resid; // largest relative mismatch at a chunk boundary that still counts as consistent
  double *out_ll, *out_logdet, *out_quad;
  int* out_status;
  // factor, only for materialising runs (replay mode 1: reference storage per
  // problem; mode 2: chunk-interleaved [problem][i][j][chunk], see replay_chunk)
  double *phi, *u, *W, *D;
};

template <int JR, int JC>
__device__ __forceinline__ void load_problem(const BatchParams& P, int b, Problem<JR, JC>& p) {
  p.load(P.a_real + (long)b * JR, P.c_real + (long)b * JR, P.a_comp + (long)b * JC,
         P.b_comp + (long)b * JC, P.c_comp + (long)b * JC, P.d_comp + (long)b * JC,
         P.jitter[b]);
}

// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() also
// drains vmcnt, which would expose the HBM latency of the staged tile prefetch (loads
// issued in step i are consumed in step i + 1) at every step.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------
// StagedSeries: the wave's 64 lanes (64 consecutive chunks of one problem) read
// the ROW-MAJOR series cooperatively in tiles of 8 steps x 64 chunks -- each load
// instruction covers 8 chunk rows x 64 contiguous bytes -- and transpose them
// through LDS (row stride 9 doubles: conflict-free ds_read_b64 by row).  The
// group g = (i+1) & 7 of the NEXT tile is requested at the beginning of step i and
// written to the other LDS buffer at the beginning of step i+1, so the HBM latency
// hides behind a whole step of arithmetic, and the per-lane reads for step i+1 are
// issued during step i.
// This replaces a separate relayout pass (0.87 ms for 3 x 0.82 GB at 5.7 TB/s).
// ---------------------------------------------------------------------------
// PRIVATE = false: the wave is its workgroup (the fences are workgroup barriers, free for a
// lone wave); PRIVATE = true: the wave shares its workgroup with other roles
// (clr_split_kernels.h) -- the tiles are still written and read by this wave only, LDS
// operations of one wave complete in order, so a compiler fence + lgkmcnt wait suffices.
template <bool PRIVATE>
struct StagedSeriesT {
  double* lds;  // [2 buffers][3 arrays][64 rows][9]
  const double *g0, *g1, *g2;  // problem bases: t, diag, y (row-major)
  long lim;                    // N
  int L, row0, nchunk, lane;
  double p0, p1, p2;  // loads in flight
  __device__ __forceinline__ void issue(int tile, int group) {
    const int r = group * 8 + (lane >> 3), col = tile * 8 + (lane & 7);
    const long n = (long)(row0 + r) * L + col;
    // a row may read up to 2 samples into the next one (t_{n+1}, t_{n+2} of its last steps)
    const bool ok = (row0 + r < nchunk) && (col < L + 2) && (n < lim);
    p0 = ok ? g0[n] : 0.0;
    p1 = ok ? g1[n] : 0.0;
    p2 = ok ? g2[n] : 0.0;
  }
  __device__ __forceinline__ void commit(int tile, int group) {
    double* d = lds + (tile & 1) * 1728 + (group * 8 + (lane >> 3)) * 9 + (lane & 7);
    d[0] = p0;
    d[576] = p1;
    d[1152] = p2;
  }
  // Schedule (tile k = local samples 8k .. 8k+7 of all 64 rows, buffer k & 1):
  //   step i reads t at i+2 and diag, y at i+1, so tile k must be complete at the top
  //   of step 8k-2.  Its group g is requested at the top of step 8(k-1) + g - 2 and
  //   written at the top of the following step (a whole step of latency hiding across
  //   the loop back-edge: if the write sits in the same step as the loads the compiler
  //   schedules it right behind them and every step stalls on HBM, +12 % measured).
  //   The prologue loads tile 0 and groups 0, 1 of tile 1.
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int gidx = 0; gidx < 8; ++gidx) {
      issue(0, gidx);
      commit(0, gidx);
    }
    issue(1, 0);
    commit(1, 0);
    issue(1, 1);
    commit(1, 1);
    if (PRIVATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
  }
  __device__ __forceinline__ void step_begin(int i) {
    if (i > 0) {
      commit(((i + 1) >> 3) + 1, (i + 1) & 7);
      if (((i + 1) & 7) == 7) {  // (not __syncthreads: keep the prefetch in flight)
        if (PRIVATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else lds_barrier();
      }
    }
    issue(((i + 2) >> 3) + 1, (i + 2) & 7);
  }
  __device__ __forceinline__ void step_end(int) {}
  __device__ __forceinline__ double rd(int a, int idx) const {
    return lds[((idx >> 3) & 1) * 1728 + a * 576 + lane * 9 + (idx & 7)];
  }
  __device__ __forceinline__ double t(int i) const { return rd(0, i); }
  __device__ __forceinline__ double diag(int i) const { return rd(1, i); }
  __device__ __forceinline__ double y(int i) const { return rd(2, i); }
};
using StagedSeries = StagedSeriesT<false>;

__device__ __forceinline__ StagedSeries make_staged(const BatchParams& P, int b, int c, double* lds) {
  StagedSeries s;
  s.lds = lds;
  s.g0 = P.t + b * P.t_stride;
  s.g1 = P.diag + b * P.diag_stride;
  s.g2 = P.y + b * P.y_stride;
  s.lim = P.N;
  s.L = P.L;
  s.row0 = blockIdx.x * 64;
  s.nchunk = P.nchunk;
  s.lane = threadIdx.x;
  s.p0 = s.p1 = s.p2 = 0.0;
  return s;
}

__device__ __forceinline__ DirectSeries make_direct(const BatchParams& P, int b, int c) {
  return DirectSeries{P.t + b * P.t_stride + c * P.lane_cs, P.diag + b * P.diag_stride + c * P.lane_cs,
                      P.y + b * P.y_stride + c * P.lane_cs, P.lane_is, P.lane_cs, P.L,
                      (long)P.N - (long)c * P.L};
}

template <int JR, int JC, bool FAST, bool STAGED>
__global__ void __launch_bounds__(64) summarize_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  __shared__ double tiles[STAGED ? 2 * 3 * 64 * 9 : 1];
  const int b = blockIdx.y;
  if (P.only_pending && P.need_scan[b] == 0) return;  // (settled by the warm path; wave-uniform)
  const int c = blockIdx.x * 64 + threadIdx.x;
  const bool store = c < P.nchunk;
  if (!STAGED && !store) return;  // (staged: every lane helps loading the tiles)
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  const long slot = (long)b * P.nchunk + (store ? c : 0);
  double ld0 = 0.0, q0 = 0.0, gamma = 0.0;
  int flag0 = 0;
  if (STAGED) {
    StagedSeries src = make_staged(P, b, c, tiles);
    summarize_chunk<JR, JC, FAST>(p, src, P.L, c * P.L, P.N, store, P.elems + slot * Wd::ELEM, &ld0,
                                  &q0, &flag0, &gamma);
  } else {
    DirectSeries src = make_direct(P, b, c);
    summarize_chunk<JR, JC, FAST>(p, src, P.L, c * P.L, P.N, store, P.elems + slot * Wd::ELEM, &ld0,
                                  &q0, &flag0, &gamma);
  }
  if (!store) return;
  if (P.cond) { P.cond[slot * 3 + 0] = gamma; P.cond[slot * 3 + 1] = 1.0; P.cond[slot * 3 + 2] = 0.0; }
  P.part[slot * 2 + 0] = ld0;
  P.part[slot * 2 + 1] = q0;
  P.flags[slot] = flag0;  // (correct_kernel raises need_exact[b] from it)
}

template <int JR, int JC>
__global__ void __launch_bounds__(64) prefix_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  P.need_exact[b] = 0;  // raised by correct_kernel
  double S[Wd::SZ], f[J];
#pragma unroll
  for (int i = 0; i < Wd::SZ; ++i) S[i] = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) f[i] = 0.0;
  for (int c = 0; c + 1 < P.nchunk; ++c) {
    double dld, dq;
    int sus;
    chunk_update<J>(P.elems + ((long)b * P.nchunk + c) * Wd::ELEM, S, f, false, true, 0.0, 0.0, &dld,
                    &dq, &sus);
    double* o = P.starts + ((long)b * P.nchunk + c + 1) * Wd::START;
#pragma unroll
    for (int i = 0; i < Wd::SZ; ++i) o[i] = S[i];
#pragma unroll
    for (int i = 0; i < J; ++i) o[Wd::SZ + i] = f[i];
  }
}

// Lane k (0..15, compile-time after unrolling) of the caller's 16-lane DPP row, for every
// lane of that row: two v_mov_b32_dpp row_newbcast.  The prefix kernel's groups ARE DPP
// rows; a __shfl is two ds_bpermute_b32 through the LDS crossbar (~100 cycles each for a
// lone wave), and the Gauss-Jordan sweep does ~90 of these broadcasts per chunk.
__device__ __forceinline__ int row_bcast_i32(int v, int k) {
#define CLR_BC(K) case K: return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, true)  /* row_newbcast:K */
  switch (k) {
    CLR_BC(0); CLR_BC(1); CLR_BC(2); CLR_BC(3); CLR_BC(4); CLR_BC(5); CLR_BC(6); CLR_BC(7);
    CLR_BC(8); CLR_BC(9); CLR_BC(10); CLR_BC(11); CLR_BC(12); CLR_BC(13); CLR_BC(14);
    default: return __builtin_amdgcn_mov_dpp(v, 0x15F, 0xf, 0xf, true);
  }
#undef CLR_BC
}
__device__ __forceinline__ double row_bcast(double v, int k) {
  return __hiloint2double(row_bcast_i32(__double2hiint(v), k), row_bcast_i32(__double2loint(v), k));
}
__device__ __forceinline__ int row_bcast_int(int v, int k) { return row_bcast_i32(v, k); }
// lane k of the caller's GROUP-lane group: DPP for 16-lane groups (= DPP rows), the LDS
// crossbar for the 32- and 64-lane groups of the wide widths (few chunks there)
template <int GROUP>
__device__ __forceinline__ double group_bcast(double v, int k) {
  if (GROUP == 16) return row_bcast(v, k);
  return __shfl(v, (threadIdx.x & ~(GROUP - 1)) + k, 64);
}
template <int GROUP>
__device__ __forceinline__ int group_bcast_int(int v, int k) {
  if (GROUP == 16) return row_bcast_i32(v, k);
  return __shfl(v, (threadIdx.x & ~(GROUP - 1)) + k, 64);
}

// ---------------------------------------------------------------------------
// Cooperative prefix: 16 lanes per problem (4 problems per wave) instead of one.
// Same algebra as the advance part of chunk_update (clr_core.h), column-per-lane:
//   lanes 0..7  of a group hold the columns of  M^T = I + P Jm
//   lanes 8..15 hold the columns of P; Gauss-Jordan on [M^T | P] turns them into
//               the columns of  G = M^-T P  (pivot row and multipliers are
//               broadcast from the pivot column's lane with wave shuffles).
//   g = M^-T h  is obtained without a 17th column from  M^-T = I - G Jm.
// Then, still one column per lane:  X = G A^T (rows, using G = G^T), an 8x8
// transpose through LDS, P' = C + A X, f' = A g + b.  The running P is kept in
// LDS (pbuf) so the matrix lanes can form M^T.  ~750 instructions per lane per
// chunk instead of ~5500, and 16x more lanes in flight: the single-lane version
// ran on 16 of the chip's 1024 SIMDs (profiles/r01b_pmc_counters.txt).
// ---------------------------------------------------------------------------
template <int J, int HALF>
__global__ void __launch_bounds__(64) prefix_coop_kernel(const BatchParams P_) {
  constexpr int GROUP = 2 * HALF, NG = 64 / GROUP;  // lanes per problem, problems per wave
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  constexpr int START = SZ + J;
  __shared__ double pbuf[NG][HALF][HALF];      // pbuf[g][j][i] = P[i][j] (column j contiguous)
  __shared__ double xbuf[NG][HALF][HALF + 1];  // transpose buffer, padded
  // the chunk's element, staged through LDS one chunk AHEAD: every lane of a group needs
  // all of A (72 load instructions per lane and chunk with their HBM latency exposed);
  // instead the group's 16 lanes fetch element c + 1 cooperatively (<= 10 coalesced loads
  // each) while chunk c is processed, and everybody reads it from LDS (group stride 162
  // doubles: the four groups' broadcast reads fall into disjoint banks)
  constexpr int ESTRIDE = ((ELEM + 15) / 16) * 16 + 2;
  constexpr int EPER = (ELEM + GROUP - 1) / GROUP;
  __shared__ double ebuf[2][NG][ESTRIDE];
  const int lane = threadIdx.x, g = lane / GROUP, l = lane % GROUP;
  const bool rhs = l >= HALF;
  const int col = l % HALF;
  const bool cv = col < J;
  const int cc = cv ? col : J - 1;
  const int prob = blockIdx.x * NG + g;
  const bool active = prob < P_.B;
  const long pb = active ? prob : P_.B - 1;
  const bool writer = rhs && cv && active;

  if (l == 0 && active) P_.need_exact[prob] = 0;  // raised by correct_kernel

  double Pc[J];  // column `col` of the running P (rhs lanes)
  double fj = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) Pc[i] = 0.0;
  if (rhs) {
#pragma unroll
    for (int i = 0; i < HALF; ++i) pbuf[g][col][i] = 0.0;
  }
  {
    const double* E0 = P_.elems + (pb * P_.nchunk) * ELEM;
#pragma unroll
    for (int m = 0; m < EPER; ++m)
      if (l + GROUP * m < ELEM) ebuf[0][g][l + GROUP * m] = E0[l + GROUP * m];
  }
  __syncthreads();

  for (int c = 0; c + 1 < P_.nchunk; ++c) {
    double nx[EPER];  // this lane's share of element c + 1, in flight during the whole iteration
    const bool more = c + 2 < P_.nchunk;
    if (more) {
      const double* En = P_.elems + (pb * P_.nchunk + c + 1) * ELEM;
#pragma unroll
      for (int m = 0; m < EPER; ++m) nx[m] = (l + GROUP * m < ELEM) ? En[l + GROUP * m] : 0.0;
    }
    const double* E = ebuf[c & 1][g];
    const double* A = E;
    const double* bv = E + J * J;
    const double* C = bv + J;
    const double* eta = C + SZ;
    const double* Jm = eta + J;

    double jc[J], et[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      jc[i] = cv ? Jm[sym(i, cc)] : 0.0;
      et[i] = eta[i];
    }
    // column of [M^T | P]
    double T[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double acc = (i == col) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc += pbuf[g][j][i] * jc[j];
      T[i] = rhs ? Pc[i] : acc;
    }
    // h = f + P eta (component `col` in rhs lane `col`), v = Jm h
    double hj = fj;
#pragma unroll
    for (int i = 0; i < J; ++i) hj += Pc[i] * et[i];
    double vj = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) vj += jc[i] * group_bcast<GROUP>(hj, HALF + i);

    // Gauss-Jordan with partial pivoting on the 2J columns of the group
#pragma unroll
    for (int c0 = 0; c0 < J; ++c0) {
      int piv = c0;
      double best = fabs(T[c0]);
#pragma unroll
      for (int i = c0 + 1; i < J; ++i) {
        const double cand = fabs(T[i]);
        const bool take = cand > best;
        best = take ? cand : best;
        piv = take ? i : piv;
      }
      piv = group_bcast_int<GROUP>(piv, c0);  // the decision of the pivot column's lane
      double top = T[c0];
      const double old_top = top;
#pragma unroll
      for (int i = c0 + 1; i < J; ++i) {
        const bool hit = (i == piv);
        top = hit ? T[i] : top;
        T[i] = hit ? old_top : T[i];
      }
      T[c0] = top;
      double m[J];
#pragma unroll
      for (int i = 0; i < J; ++i) m[i] = group_bcast<GROUP>(T[i], c0);
      const double t = T[c0] * recip_fast(m[c0]);  // (a zero / non-finite pivot gives NaN: the problem is then flagged downstream)
#pragma unroll
      for (int i = 0; i < J; ++i) T[i] = (i == c0) ? t : (T[i] - m[i] * t);
    }
    // rhs lanes: T = G[:, col] = G[col, :]
    double gj = hj;  // g = h - G v
#pragma unroll
    for (int i = 0; i < J; ++i) gj -= T[i] * group_bcast<GROUP>(vj, HALF + i);
    double fn = cv ? bv[cc] : 0.0;  // f' = A g + b
#pragma unroll
    for (int i = 0; i < J; ++i) fn += (cv ? A[cc * J + i] : 0.0) * group_bcast<GROUP>(gj, HALF + i);

    // X = G A^T: lane `col` computes row `col`; transpose through LDS to columns
    double Xr[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < J; ++i) acc += T[i] * A[j * J + i];
      Xr[j] = acc;
    }
    if (rhs) {
#pragma unroll
      for (int j = 0; j < J; ++j) xbuf[g][col][j] = Xr[j];
    }
    __syncthreads();
    double Pn[J];  // P'[:, col] = C[:, col] + A X[:, col]
#pragma unroll
    for (int k = 0; k < J; ++k) {
      double acc = cv ? C[sym(k, cc)] : 0.0;
#pragma unroll
      for (int a = 0; a < J; ++a) acc += A[k * J + a] * xbuf[g][a][col];
      Pn[k] = acc;
    }
    if (rhs) {
#pragma unroll
      for (int i = 0; i < J; ++i) {
        Pc[i] = cv ? Pn[i] : 0.0;
        pbuf[g][col][i] = Pc[i];
      }
      fj = fn;
    }
    if (writer) {
      double* o = P_.starts + (pb * P_.nchunk + c + 1) * START;
#pragma unroll
      for (int k = 0; k < J; ++k)
        if (k <= col) o[tri(k, cc)] = Pn[k];
      o[SZ + cc] = fn;
    }
    if (more) {
#pragma unroll
      for (int m = 0; m < EPER; ++m)
        if (l + GROUP * m < ELEM) ebuf[(c + 1) & 1][g][l + GROUP * m] = nx[m];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// correct: once the prefix phase has produced every chunk's start state, the
// corrections of chunk_update (true log-det / quadratic contributions from the
// zero-start sums, plus the positive-definiteness certificate) depend only on
// (start state, element) of that chunk: one lane per (problem, chunk), the
// host-checked single-lane code, ~20 us for 65 536 chunks.  This is what makes the
// replay pass unnecessary for the fused log-likelihood.
// ---------------------------------------------------------------------------
template <int J>
__global__ void __launch_bounds__(64) correct_kernel(const BatchParams P) {
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  constexpr int START = SZ + J;
  const long slot = (long)blockIdx.x * 64 + threadIdx.x;
  if (slot >= (long)P.B * P.nchunk) return;
  const int b = (int)(slot / P.nchunk), c = (int)(slot % P.nchunk);
  if (P.only_pending && P.need_scan[b] == 0) return;  // (settled by the warm path: its workspace holds nothing)
  if (P.flags[slot]) atomicOr(P.need_exact + b, 2);  // a zero-start pivot <= 0 (summarize)
  if (c == 0) {  // the first chunk starts from the zero state: nothing to correct
    if (P.egerr) P.egerr[slot] = 0.0;
    return;
  }
  double S[SZ], f[J];
  const double* st = P.starts + slot * START;
#pragma unroll
  for (int i = 0; i < SZ; ++i) S[i] = st[i];
#pragma unroll
  for (int i = 0; i < J; ++i) f[i] = st[SZ + i];
  double dld = 0.0, dq = 0.0;
  int sus = 0;
  double mu = 1.0, eg = 0.0, errq = 0.0;
  chunk_update<J>(P.elems + slot * ELEM, S, f, true, false, P.part[slot * 2 + 0], P.part[slot * 2 + 1],
                  &dld, &dq, &sus, &mu, !P.logdet_only, P.egerr ? &eg : nullptr, &errq);
  if (P.cond) { P.cond[slot * 3 + 1] = mu; P.cond[slot * 3 + 2] = errq; }  // (slot 2: until decide_kernel has read it)
  if (P.egerr) P.egerr[slot] = eg;
  P.part[slot * 2 + 0] += dld;
  P.part[slot * 2 + 1] += dq;
  if (sus) {
    P.flags[slot] |= 2;
    atomicOr(P.need_exact + b, 2);
  }
}

// FIXUP: the short second launch of a materialising run (BatchParams::fixup_steps) -- its own instantiation, so that
// profiles tell the two apart (one symbol would average a 4-ms replay with a 0.2-ms pass over the chunk heads)
template <int JR, int JC, int MATERIALIZE, bool FAST, bool STAGED, bool FIXUP = false>
__global__ void __launch_bounds__(64) replay_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  __shared__ double tiles[STAGED ? 2 * 3 * 64 * 9 : 1];
  const int b = blockIdx.y;
  // forced-exact / materialising runs replay everything; otherwise only the problems decide_kernel
  // marked ill-conditioned (level 1; level >= 2 goes straight to sequential_kernel)
  if (!P.force_exact && (P.need_exact[b] != 1 || P.defer_level1)) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  const bool mine = c < P.nchunk;
  if (!STAGED && !mine) return;
  const int n0 = c * P.L;  // >= N for lanes past the last chunk: all their steps are padding
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  double ld, qd;
  int flag;
  const long Nm1 = P.N - 1;
  constexpr bool fixup = FIXUP;
  // (a problem the sequential recurrence settled -- level >= 2 -- keeps ITS factor: wave-uniform)
  if (fixup && P.need_exact && P.need_exact[b] >= 2) return;
  if (fixup && !STAGED && c == 0) return;  // (chunk 0 starts from the exact zero state: nothing to refine)
  const double* start = (mine && c > 0) ? (fixup ? P.ends + ((long)b * P.nchunk + c - 1) * Wd::START
                                                 : P.starts + ((long)b * P.nchunk + c) * Wd::START) : nullptr;
  // (fix-up pass: lanes that only keep a staged wave's tile loads company -- chunk 0, lanes past the last chunk -- see an
  //  empty series: no sample of theirs is "valid", nothing is stored)
  const int steps = fixup ? (P.fixup_steps < P.L ? P.fixup_steps : P.L) : P.L;
  const int N_eff = (fixup && (c == 0 || !mine)) ? 0 : P.N;
  double *phi_o = nullptr, *u_o = nullptr, *W_o = nullptr, *D_o = nullptr;
  long fstride = 0;
  if (MATERIALIZE == 1) {  // reference storage, one problem after the other
    phi_o = P.phi + (long)b * J * Nm1;
    u_o = P.u + (long)b * J * Nm1;
    W_o = P.W + (long)b * J * P.N;
    D_o = P.D + (long)b * P.N;
  } else if (MATERIALIZE >= 2) {  // [problem][i][j][chunk] (3: the lean layout, W and D only)
    const long cells = (long)P.L * P.nchunk;
    const int cc = mine ? c : 0;
    fstride = P.nchunk;
    if (MATERIALIZE == 2) {
      phi_o = P.phi + (long)b * J * cells + cc;
      u_o = P.u + (long)b * J * cells + cc;
    }
    W_o = P.W + (long)b * J * cells + cc;
    D_o = P.D + (long)b * cells + cc;
  }
  double endst[Wd::START];
  if (STAGED) {
    StagedSeries src = make_staged(P, b, c, tiles);
    replay_chunk<JR, JC, MATERIALIZE, FAST>(p, src, steps, N_eff, n0, start, &ld, &qd, &flag, phi_o, u_o,
                                            W_o, D_o, fstride, endst);
  } else {
    DirectSeries src = make_direct(P, b, c);
    replay_chunk<JR, JC, MATERIALIZE, FAST>(p, src, steps, N_eff, n0, start, &ld, &qd, &flag, phi_o, u_o,
                                            W_o, D_o, fstride, endst);
  }
  if (!mine || fixup) return;  // (the fix-up pass rewrites factor entries only: sums, flags and the record stand)
  if (P.ends) {
    double* e = P.ends + ((long)b * P.nchunk + c) * Wd::START;
#pragma unroll
    for (int i = 0; i < Wd::START; ++i) e[i] = endst[i];
  }
  if (P.cond) {
    // the recurrence carried this chunk from the scanned start state of chunk c to sample (c+1) L:
    // how far is that from the scanned start state of chunk c+1?  (Both are the same quantity; the
    // difference is the rounding inconsistency of the scan algebra, measured, not estimated.)
    double res = 0.0;
    if (c + 1 < P.nchunk) {
      const double* nx = P.starts + ((long)b * P.nchunk + c + 1) * Wd::START;
      double pmax = 0.0, fmaxv = 0.0, dp = 0.0, df = 0.0;
#pragma unroll
      for (int i = 0; i < Wd::SZ; ++i) { pmax = fmax(pmax, fabs(nx[i])); dp = fmax(dp, fabs(nx[i] - endst[i])); }
#pragma unroll
      for (int i = 0; i < J; ++i) { fmaxv = fmax(fmaxv, fabs(nx[Wd::SZ + i])); df = fmax(df, fabs(nx[Wd::SZ + i] - endst[Wd::SZ + i])); }
      res = dp / pmax;  // (0/0 = NaN counts as inconsistent)
      if (!P.logdet_only && fmaxv > 0.0) res = fmax(res, df / fmaxv);
      if (!(pmax > 0.0)) res = (dp == 0.0) ? 0.0 : INFINITY;
    }
    P.cond[((long)b * P.nchunk + c) * 3 + 2] = res;
  }
  P.partx[((long)b * P.nchunk + c) * 2 + 0] = ld;
  P.partx[((long)b * P.nchunk + c) * 2 + 1] = qd;
  P.flagsx[(long)b * P.nchunk + c] = flag;
}

// The lean materialised factor of ONE problem in the reference's storage (cholesky.h:76-78, :703-706; __getstate__'s
// arrays, solver.cpp:36-42).  The replay (MATERIALIZE == 3) stored W and D only; phi[:, n] = exp(-c (t_{n+1} - t_n)) and
// u[:, n - 1] = U~(t_n) are pure functions of the times and the coefficients (cholesky.h:127-147) and are evaluated here
// by the same device functions, on the same operands, as in the replay's own step.
template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(256) expand_lean_factor_kernel(const BatchParams P, int b, const double* __restrict__ t,
                                                                 double* __restrict__ phi, double* __restrict__ u,
                                                                 double* __restrict__ W, double* __restrict__ D) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  const long cells = (long)P.L * P.nchunk;
  const double* Wi = P.W + (long)b * J * cells;
  const double* Di = P.D + (long)b * cells;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < P.N; n += gridDim.x * 256) {
    const int c = n / P.L, i = n % P.L;
    const double tn = t[n];
    D[n] = Di[(long)i * P.nchunk + c];
#pragma unroll
    for (int j = 0; j < J; ++j) W[(long)J * n + j] = Wi[((long)i * J + j) * P.nchunk + c];
    if (n >= 1) {
      double uu[J], vv[J];
      features_uv<JR, JC, FAST>(p, tn, uu, vv);
#pragma unroll
      for (int j = 0; j < J; ++j) u[(long)J * (n - 1) + j] = uu[j];
    }
    if (n + 1 < P.N) {
      double phid[nz(JR + JC)];
      features_phi_distinct<JR, JC>(p, t[n + 1] - tn, phid);
#pragma unroll
      for (int j = 0; j < J; ++j) phi[(long)J * n + j] = phid[phi_index<JR>(j)];
    }
  }
}

// decide: after correct_kernel, one wave per problem: a problem the certificate did not flag but
// whose conditioning record says the chunk summaries cannot be trusted to 1e-10 leaves the replay-free
// route (level 1: chunked replay + end-state check).  Round 3 calibration (profiles/r03_conditioning_
// calibration.txt: 5400 adversarial problem x chunking combinations, widths 1..8 and 32): the deviation
// from the oracle follows gamma = max a_n / D_n -- the cancellation in the recurrence ITSELF, which any
// implementation shares (log-log correlation 0.90; with 1 / mu 0.64, with gamma / mu 0.85) -- so the
// record is tested as gamma_max < cert_gamma_abs (1e4) AND gamma_max / mu_min < cert_gamma (1e7): more
// problems settled from the summaries than with round 2's gamma / mu < 1e6, at a smaller worst deviation
// (2.4e-12 against 5e-11), and BASELINE config 4 (gamma ~ 1.3e3, mu 8e-4 .. 2e-2) no longer sends a sixth
// of its problems through 7 ms of checked replay.  A third test replaces the pessimistic 1 / mu by a MEASUREMENT:
// the correct kernels compute the first-order forward error eG of every chunk's G = (I + P Jm)^-1 P from its
// residual (chunk_update, eg_out) and the problem must have gamma_max x eG_max < cert_eg (3e-9); this also catches
// the one outlier of the calibration set that both gamma tests let through (a 6-chunk N = 50 problem, 1.9e-9):
// with all three, 2166 of 3488 settled from the summaries, worst 1.05e-11 (round 2: 2109, worst 1.9e-9).
template <int J>
__global__ void __launch_bounds__(256) decide_kernel(const BatchParams P) {
  // one workgroup per problem (64 threads, or 256 for the thousands of chunks of one long series): max / min over the
  // chunks' records are order-independent, the sums only feed a threshold
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
  if (P.only_pending && P.need_scan[b] == 0) return;
  if (!P.cond) {
    // no conditioning record: nothing vouches for the chunk summaries (the correct kernels leave their rounding-error
    // estimates in it) -- the problem takes the checked replay instead of being settled silently
    if (tid == 0 && P.need_exact[b] == 0) P.need_exact[b] = 1;
    return;
  }
  // NaN records stick (and then fail the comparison below: ill-conditioned)
  auto nmax = [](double a, double x) { return (a != a) ? a : ((x != x) ? x : (x > a ? x : a)); };
  auto nmin = [](double a, double x) { return (a != a) ? a : ((x != x) ? x : (x < a ? x : a)); };
  double g = 0.0, m = 1.0, e = 0.0;
  // the corrections' rounding-error estimates (chunk_update: J eps / mu, times |w.G w| for the quadratic form) summed
  // over the chunks, against the problem's own log det and quadratic form
  double err_ld = 0.0, err_q = 0.0, sum_ld = 0.0, sum_q = 0.0;
  for (int c = tid; c < P.nchunk; c += NT) {
    const long slot = (long)b * P.nchunk + c;
    g = nmax(g, P.cond[slot * 3]);
    const double mu = P.cond[slot * 3 + 1];
    m = nmin(m, mu);
    if (P.egerr) e = nmax(e, P.egerr[slot]);
    if (c > 0) err_ld += J * 2.2e-16 / mu;
    err_q += P.cond[slot * 3 + 2];
    P.cond[slot * 3 + 2] = 0.0;  // (from here on: the replay's end-state residual)
    sum_ld += P.part[slot * 2 + 0];
    sum_q += P.part[slot * 2 + 1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    g = nmax(g, __shfl_xor(g, off, 64));
    m = nmin(m, __shfl_xor(m, off, 64));
    e = nmax(e, __shfl_xor(e, off, 64));
    err_ld += __shfl_xor(err_ld, off, 64);
    err_q += __shfl_xor(err_q, off, 64);
    sum_ld += __shfl_xor(sum_ld, off, 64);
    sum_q += __shfl_xor(sum_q, off, 64);
  }
  if (NT > 64) {  // (wave-uniform) the other waves' results through LDS
    __shared__ double red[4][7];
    if (lane == 0) {
      double* o = red[tid >> 6];
      o[0] = g; o[1] = m; o[2] = e; o[3] = err_ld; o[4] = err_q; o[5] = sum_ld; o[6] = sum_q;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NT / 64; ++w) {
        g = nmax(g, red[w][0]); m = nmin(m, red[w][1]); e = nmax(e, red[w][2]);
        err_ld += red[w][3]; err_q += red[w][4]; sum_ld += red[w][5]; sum_q += red[w][6];
      }
    }
  }
  if (tid != 0 || P.need_exact[b] != 0) return;
  bool replay = !(err_ld <= 3e-12 * fabs(sum_ld)) || (!P.logdet_only && !(err_q <= 3e-12 * fabs(sum_q)));
  // (NaN records count as ill-conditioned)
  if (P.cert_gamma > 0.0 && (!(g < P.cert_gamma * m) || (P.cert_gamma_abs > 0.0 && !(g < P.cert_gamma_abs)) ||
                             (P.cert_eg > 0.0 && P.egerr && !(g * e < P.cert_eg))))
    replay = true;
  if (replay) P.need_exact[b] = 1;
}

// After the chunked replay (level 1 and forced-exact / materialising runs): trust it iff every chunk's end state met the
// scanned start state of the next chunk (cond[.][2] <= cert_resid), else the problem goes to the sequential recurrence
// (level 2).  One wave per problem, lanes striding over its chunks (round 4: one long series has thousands of chunks --
// as a loop of the sequential kernel's lone lane the 4224 dependent loads took 0.47 ms of a 1.1 ms call).
template <int J>
__global__ void __launch_bounds__(64) check_replay_kernel(const BatchParams P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (!P.cond) return;
  if (P.only_pending && P.need_scan[b] == 0) return;
  const int level = P.need_exact[b];
  if (level >= 2 || !(level == 1 || P.force_exact)) return;
  if (P.defer_level1 && !P.force_exact) return;  // (no replay ran for it: pending, see finalize_kernel)
  double r = 0.0;
  for (int c = lane; c < P.nchunk; c += 64) {
    const double rc = P.cond[((long)b * P.nchunk + c) * 3 + 2];
    r = (rc != rc) ? INFINITY : fmax(r, rc);  // (a NaN residual counts as inconsistent)
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) r = fmax(r, __shfl_xor(r, off, 64));
  if (lane == 0 && !(r <= P.cert_resid)) P.need_exact[b] = 2;
}

// ---------------------------------------------------------------------------
// sequential: the reference recurrence, truly sequential -- one lane per problem walks its
// chunks in order and carries the state in registers, so nothing depends on the scanned
// start states.  It settles (a) problems with a chunk the certificate could not settle
// (need_exact raised by correct_kernel) and (b) problems whose conditioning record says the
// scanned start states cannot be trusted to 1e-10: gamma_max / mu_min >= cert_gamma, where
// gamma = a_n / D_n measures the cancellation in the pivots and 1 / mu the conditioning of
// the chunk corrections (calibrated on tests/_cases.adversarial, profiles/r02k_adv_probe.txt:
// below 1e6 the scan stays within 2.8e-11 of the oracle, above it reaches 1e-6).  The chunked
// replay from scanned starts inherits their error: it is only used to write the
// factor of well-conditioned problems; a flagged problem's factor columns are rewritten here.
// Outputs are per-chunk partials in partx / flagsx exactly like replay_kernel's.
// ---------------------------------------------------------------------------
template <int JR, int JC, int MATERIALIZE, bool FAST>
__global__ void __launch_bounds__(64) sequential_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  constexpr int J = Wd::J;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  int level = P.need_exact[b];
  if (P.nchunk <= 256 && level < 2 && (level == 1 || P.force_exact) && P.cond && !(P.defer_level1 && !P.force_exact)) {
    // the chunked replay ran for this problem: trust it iff every chunk's end state met the scanned start state of the
    // next chunk (longer chunk lists: check_replay_kernel has done this, a wave per problem)
    double r = 0.0;
    for (int c = 0; c < P.nchunk; ++c) {
      const double rc = P.cond[((long)b * P.nchunk + c) * 3 + 2];
      r = (rc != rc) ? INFINITY : fmax(r, rc);  // (a NaN residual counts as inconsistent -- and stays so)
    }
    if (!(r <= P.cert_resid)) level = 2;
  }
  if (level < 2) return;
  P.need_exact[b] = 2;
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  double state[Wd::START];
  const long Nm1 = P.N - 1;
  for (int c = 0; c < P.nchunk; ++c) {
    DirectSeries src = make_direct(P, b, c);
    double ld, qd;
    int flag;
    double *phi_o = nullptr, *u_o = nullptr, *W_o = nullptr, *D_o = nullptr;
    long fstride = 0;
    if (MATERIALIZE == 1) {
      phi_o = P.phi + (long)b * J * Nm1;
      u_o = P.u + (long)b * J * Nm1;
      W_o = P.W + (long)b * J * P.N;
      D_o = P.D + (long)b * P.N;
    } else if (MATERIALIZE >= 2) {
      const long cells = (long)P.L * P.nchunk;
      fstride = P.nchunk;
      if (MATERIALIZE == 2) {
        phi_o = P.phi + (long)b * J * cells + c;
        u_o = P.u + (long)b * J * cells + c;
      }
      W_o = P.W + (long)b * J * cells + c;
      D_o = P.D + (long)b * cells + c;
    }
    replay_chunk<JR, JC, MATERIALIZE, FAST>(p, src, P.L, P.N, c * P.L, c > 0 ? state : nullptr, &ld, &qd, &flag,
                                            phi_o, u_o, W_o, D_o, fstride, state);
    const long slot = (long)b * P.nchunk + c;
    P.partx[slot * 2 + 0] = ld;
    P.partx[slot * 2 + 1] = qd;
    P.flagsx[slot] = flag;
  }
}

// ---------------------------------------------------------------------------
// Warm-started plain recurrence: the scan's riders (A, eta, Jm), its prefix and its corrections exist because a
// chunk's start state depends on everything before it.  On series that FORGET -- the decay between samples is not
// small, e.g. the paper's accuracy family (paper/figures/error/error.py:24-25: mean spacing 0.8, c >= 1): after a
// few tens of samples the state no longer depends on where it started -- they are pure overhead: every chunk runs
// the reference recurrence itself (replay_chunk: ~1x the reference's arithmetic, two waves per SIMD, reading a padded
// chunk-interleaved copy of the series that carries every chunk's warm-up rows: relayout_warm_kernel) from the ZERO
// state K samples before its first sample, counts nothing during those K warm-up steps, and the state it has
// reached at its first sample is compared with the state the PREVIOUS chunk reaches at that very sample
// (warm_check_kernel).  Chunk 0 starts from the true zero state, so agreement at every boundary certifies all
// start states by induction; a problem with a mismatch above warm_resid, a flagged pivot or a non-finite partial
// is left to the scan pipeline (need_scan).  K is chosen per problem on the host from the slowest decay rate and
// the time the K samples before every chunk boundary span (api_batch.hip); the check makes that choice safe, not just
// plausible.  No prefix, no corrections, no LDS.
// ---------------------------------------------------------------------------
// The warm kernel's view of the series: a lane walks down ITS column of the warm copy (one coalesced 512-B wave
// load per array and step), rows are prefetched PF steps ahead into registers, every load is unguarded (the copy is
// padded), the offset is one running wave-uniform value.  No LDS: the kernel's occupancy is set by its registers
// alone (two waves per SIMD; the LDS-staged tiles of StagedSeries would allow five waves per CU).
struct WarmQueueSeries {
  static constexpr int PF = 4;
  const double *tp, *dp, *yp;  // the lane's column, at the row of its first warm-up step
  long stride;                 // doubles between rows (= chunks of the warm path)
  long o;                      // offset of the next row to fetch for diag / y (t: one row further)
  double tq[PF], dq[PF], yq[PF];  // at step i: tq[k] = t(i + 2 + k), dq[k] = diag(i + 1 + k), yq[k] = y(i + 1 + k)
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      tq[k] = tp[(2 + k) * stride];
      dq[k] = dp[(1 + k) * stride];
      yq[k] = yp[(1 + k) * stride];
    }
    o = (1 + PF) * stride;
  }
  // (replay_chunk asks for t(0), t(1), diag(0), y(0) once, then for t(i + 2), diag(i + 1), y(i + 1) at step i)
  __device__ __forceinline__ double t(int idx) const { return idx < 2 ? tp[idx * stride] : tq[0]; }
  __device__ __forceinline__ double diag(int idx) const { return idx < 1 ? dp[0] : dq[0]; }
  __device__ __forceinline__ double y(int idx) const { return idx < 1 ? yp[0] : yq[0]; }
  __device__ __forceinline__ void step_begin(int) {}
  __device__ __forceinline__ void step_end(int) {
#pragma unroll
    for (int k = 0; k + 1 < PF; ++k) { tq[k] = tq[k + 1]; dq[k] = dq[k + 1]; yq[k] = yq[k + 1]; }
    tq[PF - 1] = tp[o + stride];
    dq[PF - 1] = dp[o];
    yq[PF - 1] = yp[o];
    o += stride;
  }
};

template <int JR, int JC, bool FAST>
__global__ void __launch_bounds__(64) warm_kernel(const BatchParams P) {
  using Wd = Widths<JR, JC>;
  const int b = blockIdx.y;
  const int K = P.wK[b];
  if (K <= 0) return;  // this problem takes the scan (wave-uniform)
  const int c = blockIdx.x * 64 + threadIdx.x;
  const bool mine = c < P.wnchunk;
  const int cc = mine ? c : P.wnchunk - 1;  // (lanes past the last chunk recompute it and store nothing)
  Problem<JR, JC> p;
  load_problem<JR, JC>(P, b, p);
  WarmQueueSeries src;
  const long first = (long)(P.wKpad - K) * P.wnchunk + cc;
  src.tp = P.wt + b * P.wt_stride + first;
  src.dp = P.wdiag + b * P.wdiag_stride + first;
  src.yp = P.wy + b * P.wy_stride + first;
  src.stride = P.wnchunk;
  const long slot = (long)b * P.wnchunk + cc;
  double ld, qd;
  int flag;
  double endst[Wd::START];
  replay_chunk<JR, JC, 0, FAST>(p, src, P.wL + K, P.N, cc * P.wL - K, nullptr, &ld, &qd, &flag, nullptr, nullptr,
                                nullptr, nullptr, 0, endst, K, mine ? P.wstarts + slot * Wd::START : nullptr);
  if (!mine) return;
  double* e = P.wends + slot * Wd::START;
#pragma unroll
  for (int i = 0; i < Wd::START; ++i) e[i] = endst[i];
  P.wpart[slot * 2 + 0] = ld;
  P.wpart[slot * 2 + 1] = qd;
  P.wflags[slot] = flag;
}

// One wave per problem: every boundary's two versions of the same state must agree; then the partial sums are added
// in a fixed order (as finalize_kernel) and the problem is done.  Otherwise need_scan[b] = 1.
template <int J>
__global__ void __launch_bounds__(64) warm_check_kernel(const BatchParams P) {
  constexpr int SZ = J * (J + 1) / 2, START = SZ + J;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (P.wK[b] <= 0) {
    if (lane == 0) { P.need_scan[b] = 1; P.out_status[b] = CLR_PENDING_STATUS; if (P.wresid) P.wresid[b] = 0.0; }
    return;
  }
  double res = 0.0, ld = 0.0, qd = 0.0;
  int bad = 0;
  for (int c = lane; c < P.wnchunk; c += 64) {
    const long slot = (long)b * P.wnchunk + c;
    const double l = P.wpart[slot * 2 + 0], q = P.wpart[slot * 2 + 1];
    ld += l;
    qd += q;
    if (P.wflags[slot] || !isfinite(l) || (!P.logdet_only && !isfinite(q))) bad = 1;
    if (c + 1 < P.wnchunk) {
      const double* en = P.wends + slot * START;
      const double* st = P.wstarts + (slot + 1) * START;
      double pm = 0.0, dp = 0.0, fm = 0.0, df = 0.0;
#pragma unroll
      for (int i = 0; i < SZ; ++i) { pm = fmax(pm, fabs(en[i])); dp = fmax(dp, fabs(en[i] - st[i])); }
#pragma unroll
      for (int i = 0; i < J; ++i) { fm = fmax(fm, fabs(en[SZ + i])); df = fmax(df, fabs(en[SZ + i] - st[SZ + i])); }
      double r = (pm > 0.0) ? dp / pm : (dp == 0.0 ? 0.0 : INFINITY);
      if (!P.logdet_only && fm > 0.0) r = fmax(r, df / fm);
      res = (res != res) ? res : ((r != r) ? r : fmax(res, r));  // (NaN sticks)
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(res, off, 64);
    res = (res != res || o != o) ? NAN : fmax(res, o);
    ld += __shfl_xor(ld, off, 64);
    qd += __shfl_xor(qd, off, 64);
  }
  bad = __any(bad) ? 1 : 0;
  if (lane != 0) return;
  if (P.wresid) P.wresid[b] = res;
  if (bad || !(res <= P.warm_resid)) {
    P.need_scan[b] = 1;
    P.out_status[b] = CLR_PENDING_STATUS;  // the host runs the scan pipeline for it before handing results out
    return;
  }
  P.need_scan[b] = 0;
  P.need_exact[b] = 0;
  P.out_status[b] = CLR_OK_STATUS;
  P.out_logdet[b] = ld;
  P.out_quad[b] = qd;
  P.out_ll[b] = combine_loglike(ld, qd, P.N);
}

}  // namespace clr
#include "clr_prefix_kernels.h"
#include "clr_grad_kernels.h"
#include "clr_bsolve_kernels.h"
#include "clr_bdotl_kernels.h"
#include "clr_bdot_kernels.h"
namespace clr {

// One table entry per (JR, JC): host-callable launchers.
struct BatchLaunchers {
  void (*summarize)(const BatchParams&, hipStream_t);  // (role-split kernel when P.split > 0, widths 7, 8)
  void (*prefix)(const BatchParams&, hipStream_t);
  void (*correct)(const BatchParams&, hipStream_t);
  void (*replay)(const BatchParams&, int materialize, hipStream_t);  // 0 none, 1 reference, 2 interleaved, 3 interleaved lean (W, D)
  void (*sequential)(const BatchParams&, int materialize, hipStream_t);
  // cross-check: compose the chunk elements in groups of g with the cooperative kernel (-> coop) and with the
  // single-lane host-checked form (-> ref); both [B][ceil(nchunk / g)][ELEM]
  void (*compose_check)(const BatchParams&, int g, double* coop, double* ref, hipStream_t);
  void (*warm)(const BatchParams&, hipStream_t);  // warm_kernel + warm_check_kernel
  void (*grad)(const BatchParams&, hipStream_t);  // riders + tangents + walk over the chunks (needs P.fast_trig)
  void (*grad_reverse)(const BatchParams&, hipStream_t);  // riders + record, adjoint walk, reverse sweep, reduction
  // K^-1 b for all problems and right-hand sides from the materialised factor (clr_bsolve_kernels.h): S.xT in / out
  void (*bsolve)(const BatchParams&, BSolveParams S, hipStream_t);
  // L z for all problems and right-hand sides from the materialised factor (clr_bdotl_kernels.h): S.xT in / out
  void (*bdotl)(const BatchParams&, BDotLParams S, hipStream_t);
  // K z for all problems and right-hand sides from the plan's times and coefficients (clr_bdot_kernels.h): S.zT -> S.yT
  void (*bdot)(const BatchParams&, BDotParams S, hipStream_t);
  // lean factor of problem b (replay mode 3) -> the reference's storage, phi and u regenerated (t: the problem's row-major times)
  void (*expand)(const BatchParams&, int b, const double* t, double* phi, double* u, double* W, double* D, hipStream_t);
  int elem_doubles, start_doubles;
};

bool launch_summarize_split(const BatchParams& P, int JR, int JC, hipStream_t s);
template <int JR, int JC>
struct BatchImpl {
  static void summarize(const BatchParams& P, hipStream_t s) {
    if (P.split && !P.staged && launch_summarize_split(P, JR, JC, s)) return;
    dim3 grid((P.nchunk + 63) / 64, P.B);
#define CLR_GO(F, S) hipLaunchKernelGGL((summarize_kernel<JR, JC, F, S>), grid, dim3(64), 0, s, P)
    if (P.fast_trig) { if (P.staged) CLR_GO(true, true); else CLR_GO(true, false); }
    else             { if (P.staged) CLR_GO(false, true); else CLR_GO(false, false); }
#undef CLR_GO
  }
  static void prefix(const BatchParams& P, hipStream_t s) {
    if (P.nchunk < 2) return;
    if (P.coop_prefix == 2 && P.plan.levels > 0 && P.lvl_elems && P.lvl_starts)
      launch_multilevel_prefix<JR + 2 * JC>(P, s);
    else if (P.coop_prefix)
      hipLaunchKernelGGL((prefix_coop_kernel<JR + 2 * JC, 8>), dim3((P.B + 3) / 4), dim3(64), 0, s, P);
    else
      hipLaunchKernelGGL((prefix_kernel<JR, JC>), dim3((P.B + 63) / 64), dim3(64), 0, s, P);
  }
  static void correct(const BatchParams& P, hipStream_t s) {
    if (P.nchunk < 2) return;
    const long lanes = (long)P.B * P.nchunk;
    hipLaunchKernelGGL((correct_kernel<JR + 2 * JC>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s,
                       P);
    hipLaunchKernelGGL((decide_kernel<JR + 2 * JC>), dim3(P.B), dim3(P.nchunk > 256 ? 256 : 64), 0, s, P);
  }
  static void replay(const BatchParams& P, int materialize, hipStream_t s) {
    dim3 grid((P.nchunk + 63) / 64, P.B);
#define CLR_GO(M, F, S) hipLaunchKernelGGL((replay_kernel<JR, JC, M, F, S>), grid, dim3(64), 0, s, P)
#define CLR_GOF(M, F, S) hipLaunchKernelGGL((replay_kernel<JR, JC, M, F, S, true>), grid, dim3(64), 0, s, P)
#define CLR_GO2(M)                                                                \
  if (P.fast_trig) { if (P.staged) CLR_GO(M, true, true); else CLR_GO(M, true, false); } \
  else             { if (P.staged) CLR_GO(M, false, true); else CLR_GO(M, false, false); }
#define CLR_GOF2(M)                                                                \
  if (P.fast_trig) { if (P.staged) CLR_GOF(M, true, true); else CLR_GOF(M, true, false); } \
  else             { if (P.staged) CLR_GOF(M, false, true); else CLR_GOF(M, false, false); }
    if (P.fixup_steps > 0) {  // (the chunk heads of a materialising run; mode 1: the object API's lazy pass, api_solver.hip)
      if (materialize == 3) { CLR_GOF2(3) } else if (materialize == 2) { CLR_GOF2(2) } else if (materialize == 1) { CLR_GOF2(1) }
      return;
    }
    if (materialize == 3) { CLR_GO2(3) } else if (materialize == 2) { CLR_GO2(2) } else if (materialize == 1) { CLR_GO2(1) } else { CLR_GO2(0) }
#undef CLR_GOF2
#undef CLR_GO2
#undef CLR_GOF
#undef CLR_GO
  }
  static void sequential(const BatchParams& P, int materialize, hipStream_t s) {
    if (P.nchunk < 2) return;  // (one chunk: the replay from the zero state IS the recurrence)
    dim3 grid((P.B + 63) / 64);
    if (P.nchunk > 256) hipLaunchKernelGGL((check_replay_kernel<JR + 2 * JC>), dim3(P.B), dim3(64), 0, s, P);  // (else: sequential_kernel's own loop)
#define CLR_GO(M, F) hipLaunchKernelGGL((sequential_kernel<JR, JC, M, F>), grid, dim3(64), 0, s, P)
#define CLR_GO2(M) if (P.fast_trig) CLR_GO(M, true); else CLR_GO(M, false);
    if (materialize == 3) { CLR_GO2(3) } else if (materialize == 2) { CLR_GO2(2) } else if (materialize == 1) { CLR_GO2(1) } else { CLR_GO2(0) }
#undef CLR_GO2
#undef CLR_GO
  }
  // one problem's LEAN factor (W, D chunk-interleaved) -> the reference's four arrays: W and D de-interleaved, phi and u
  // regenerated from the times and the coefficients with the very device functions the replay evaluates them with
  static void expand(const BatchParams& P, int b, const double* t_rowmajor, double* phi, double* u, double* W, double* D,
                     hipStream_t s) {
    const int blocks = std::min((P.N + 255) / 256, 4096);
    if (P.fast_trig) hipLaunchKernelGGL((expand_lean_factor_kernel<JR, JC, true>), dim3(blocks), dim3(256), 0, s, P, b, t_rowmajor, phi, u, W, D);
    else hipLaunchKernelGGL((expand_lean_factor_kernel<JR, JC, false>), dim3(blocks), dim3(256), 0, s, P, b, t_rowmajor, phi, u, W, D);
  }
  // the batched solve: chunk maps once, then per right-hand side the forward and the backward chunked affine scans
  template <bool LEAN, bool FAST>
  static void bsolve_go(const BatchParams& P, BSolveParams S, hipStream_t s) {
    constexpr int J = JR + 2 * JC;
    const dim3 grid((P.nchunk + 63) / 64, P.B), pgrid((unsigned)(((long)P.B * S.nrhs + 63) / 64));
    for (int r = 0; r < S.nrhs; ++r) {
      S.r = r;
      if (r == 0 && !S.have_M) hipLaunchKernelGGL((bsolve_summarize_kernel<JR, JC, LEAN, FAST, true>), grid, dim3(64), 0, s, P, S);
      else hipLaunchKernelGGL((bsolve_summarize_kernel<JR, JC, LEAN, FAST, false>), grid, dim3(64), 0, s, P, S);
    }
    hipLaunchKernelGGL((bsolve_prefix_kernel<J, false>), pgrid, dim3(64), 0, s, P, S);
    for (int r = 0; r < S.nrhs; ++r) {
      S.r = r;
      hipLaunchKernelGGL((bsolve_forward_kernel<JR, JC, LEAN, FAST>), grid, dim3(64), 0, s, P, S);
    }
    for (int r = 0; r < S.nrhs; ++r) {
      S.r = r;
      hipLaunchKernelGGL((bsolve_backward_kernel<JR, JC, LEAN, FAST, true>), grid, dim3(64), 0, s, P, S);
    }
    hipLaunchKernelGGL((bsolve_prefix_kernel<J, true>), pgrid, dim3(64), 0, s, P, S);
    for (int r = 0; r < S.nrhs; ++r) {
      S.r = r;
      hipLaunchKernelGGL((bsolve_backward_kernel<JR, JC, LEAN, FAST, false>), grid, dim3(64), 0, s, P, S);
    }
  }
  static void bsolve(const BatchParams& P, BSolveParams S, hipStream_t s) {
    if (S.lean) { if (P.fast_trig) bsolve_go<true, true>(P, S, s); else bsolve_go<true, false>(P, S, s); }
    else bsolve_go<false, true>(P, S, s);  // (the stored phi, u: no trigonometry)
  }
  // the batched dot_L: chunk offsets and decay products, the walk over the chunks, the recurrence from the start states
  template <bool LEAN, bool FAST>
  static void bdotl_go(const BatchParams& P, const BDotLParams& S, hipStream_t s) {
    const dim3 grid((P.nchunk + 63) / 64, P.B, S.nrhs), pgrid((unsigned)(((long)P.B * S.nrhs + 63) / 64));
    hipLaunchKernelGGL((bdotl_kernel<JR, JC, LEAN, FAST, false>), grid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdotl_prefix_kernel<JR + 2 * JC>), pgrid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdotl_kernel<JR, JC, LEAN, FAST, true>), grid, dim3(64), 0, s, P, S);
  }
  static void bdotl(const BatchParams& P, BDotLParams S, hipStream_t s) {
    if (S.lean) { if (P.fast_trig) bdotl_go<true, true>(P, S, s); else bdotl_go<true, false>(P, S, s); }
    else bdotl_go<false, true>(P, S, s);  // (the stored phi, u: no trigonometry)
  }
  // the batched dot: per triangle the chunk offsets (and, once, the decay products), the walk, the recurrence
  template <bool FAST>
  static void bdot_go(const BatchParams& P, const BDotParams& S, hipStream_t s) {
    constexpr int J = JR + 2 * JC;
    const dim3 grid((P.nchunk + 63) / 64, P.B, S.nrhs), pgrid((unsigned)(((long)P.B * S.nrhs + 63) / 64));
    hipLaunchKernelGGL((bdot_kernel<JR, JC, FAST, 0, false>), grid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdot_prefix_kernel<J, 0>), pgrid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdot_kernel<JR, JC, FAST, 0, true>), grid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdot_kernel<JR, JC, FAST, 1, false>), grid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdot_prefix_kernel<J, 1>), pgrid, dim3(64), 0, s, P, S);
    hipLaunchKernelGGL((bdot_kernel<JR, JC, FAST, 1, true>), grid, dim3(64), 0, s, P, S);
  }
  static void bdot(const BatchParams& P, BDotParams S, hipStream_t s) {
    if (P.fast_trig) bdot_go<true>(P, S, s); else bdot_go<false>(P, S, s);
  }
  static void compose_check(const BatchParams& P, int g, double* coop, double* ref, hipStream_t s) {
    constexpr int J = JR + 2 * JC;
    const int np = (P.nchunk + g - 1) / g;
    SegParams S{P.elems, nullptr, nullptr, coop, P.B, P.nchunk, g, np, nullptr};
    const long nseg = (long)P.B * np;
    hipLaunchKernelGGL((group_compose_kernel<J>), dim3((unsigned)((nseg + 1) / 2)), dim3(64), 0, s, S);
    S.parents = ref;
    hipLaunchKernelGGL((group_compose_reference_kernel<J>), dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, s, S);
  }
  static void warm(const BatchParams& P, hipStream_t s) {
    dim3 grid((P.wnchunk + 63) / 64, P.B);
    if (P.fast_trig) hipLaunchKernelGGL((warm_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    else hipLaunchKernelGGL((warm_kernel<JR, JC, false>), grid, dim3(64), 0, s, P);
    hipLaunchKernelGGL((warm_check_kernel<JR + 2 * JC>), dim3(P.B), dim3(64), 0, s, P);
  }
  static void grad(const BatchParams& P, hipStream_t s) {
    using Sh = GradShape<JR, JC>;
    const dim3 grid((P.g_nchunk + 63) / 64, P.B);
    hipLaunchKernelGGL((grad_riders_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    hipLaunchKernelGGL((grad_tangent_kernel<JR, JC, true>), dim3(grid.x, P.B, Sh::GROUPS), dim3(64), 0, s, P);
    const long n = (long)P.B * Sh::NG;
    hipLaunchKernelGGL((grad_combine_kernel<JR + 2 * JC>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, P, Sh::NG);
  }
  static void grad_reverse(const BatchParams& P, hipStream_t s) {
    using Sh = GradShape<JR, JC>;
    const dim3 grid((P.g_nchunk + 63) / 64, P.B);
    if (P.g_from_elems) {  // (a gradient chunk is a scan chunk: riders from the scan's elements, record by the plain recurrence)
      hipLaunchKernelGGL((grad_riders_elem_kernel<JR + 2 * JC>), grid, dim3(64), 0, s, P);
      hipLaunchKernelGGL((grad_record_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    } else {
      hipLaunchKernelGGL((grad_riders_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    }
    constexpr int JW = JR + 2 * JC;
    if (P.g_seg > 0 && P.g_grp_riders) {  // (one long series: thousands of gradient chunks)
      const int ngr = (P.g_nchunk + P.g_seg - 1) / P.g_seg;
      hipLaunchKernelGGL((grad_riders_compose_kernel<JW>), dim3(ngr, P.B), dim3(64), 0, s, P.g_riders, P.g_grp_riders,
                         P.need_exact, P.g_nchunk, P.g_seg);
      hipLaunchKernelGGL((grad_adjoint_kernel<JW>), dim3(1, P.B), dim3(64), 0, s,
                         AdjointWalk{P.g_grp_riders, nullptr, P.g_grp_adj, P.need_exact, ngr, ngr});
      hipLaunchKernelGGL((grad_adjoint_kernel<JW>), dim3(ngr, P.B), dim3(64), 0, s,
                         AdjointWalk{P.g_riders, P.g_grp_adj, P.g_adj, P.need_exact, P.g_nchunk, P.g_seg});
    } else {
      hipLaunchKernelGGL((grad_adjoint_kernel<JW>), dim3(1, P.B), dim3(64), 0, s,
                         AdjointWalk{P.g_riders, nullptr, P.g_adj, P.need_exact, P.g_nchunk, P.g_nchunk});
    }
    hipLaunchKernelGGL((grad_backward_kernel<JR, JC, true>), grid, dim3(64), 0, s, P);
    const int nslab = (P.g_nchunk + 255) / 256;
    hipLaunchKernelGGL((grad_reduce_slab_kernel<JR + 2 * JC>), dim3(nslab, P.B), dim3(256), 0, s, P, Sh::NG);
    hipLaunchKernelGGL((grad_reduce_kernel<JR + 2 * JC>), dim3(P.B), dim3(64), 0, s, P, Sh::NG, nslab);
  }
  static BatchLaunchers table() {
    return BatchLaunchers{&summarize, &prefix, &correct, &replay, &sequential, &compose_check, &warm, &grad, &grad_reverse,
                          &bsolve, &bdotl, &bdot, &expand, Widths<JR, JC>::ELEM, Widths<JR, JC>::START};
  }
};

// Role-split summarize (clr_split_kernels.h; batch_split*.hip): false when (JR, JC, P.split) has no
// instantiation (widths below 7) -- the caller then runs the single-wave kernel.
bool launch_summarize_split(const BatchParams& P, int JR, int JC, hipStream_t s);
bool have_summarize_split(int JR, int JC);

// the wide scan's prefix AND corrections in one walk per problem (wide64_kernels.hip): the padded width 64; 32 as a
// cross-check of the two-kernel path.  Non-zero: the kernel could not be configured (LDS).
int launch_wide_walk(const BatchParams& P, int width_padded, hipStream_t s);
// the padded width of the wide kernels' chunk algebra for a total width W (9..64)
inline int wide_padded_width(int W) { return W <= 16 ? 16 : (W <= 32 ? 32 : 64); }
// decide_kernel at the padded widths of the wide scan
void launch_wide_decide(const BatchParams& P, int width_padded, hipStream_t s);
void launch_wide_check_replay(const BatchParams& P, hipStream_t s);
void launch_wide_head_decide(const BatchParams& P, hipStream_t s);
// Per-problem reduction of the chunk partials + the -inf rules (api_kernels.hip).
void launch_finalize(const BatchParams& P, hipStream_t s);
// One problem's interleaved factor -> the reference's storage (api_kernels.hip).
void launch_deinterleave_factor(const double* phi_i, const double* u_i, const double* W_i,
                                const double* D_i, double* phi, double* u, double* W, double* D,
                                int N, int J, int L, int nchunk, hipStream_t s);
// ... and back: [problem][i][chunk] -> [problem][n] (api_kernels.hip)
void launch_relayout_back(const double* src, long src_stride, double* dst, long dst_stride, int nsrc, int N, int L, int nchunk,
                          hipStream_t s);
// [problem][n] -> [problem][i][chunk] for n = chunk * L + i (api_kernels.hip).
void launch_relayout(const double* src, long src_stride, double* dst, long dst_stride, int nsrc,
                     int N, int L, int nchunk, int pad_kind, hipStream_t s);
// [problem][n] -> the warm kernel's [problem][row][chunk], row r of chunk c = sample c L - Kpad + r (api_kernels.hip)
void launch_relayout_warm(const double* src, long src_stride, double* dst, long dst_stride, int nsrc, int N, int L,
                          int nchunk, int Kpad, int rows, int pad_kind, hipStream_t s);

// a few problems of a plan re-planned as a small plan of their own (api_kernels.hip; api_batch.hip: rescue_run)
void launch_gather_series(const double* src, long src_stride, double* dst, const int* idx, int n, int N, hipStream_t s);
void launch_gather_coeffs(const double* src, double* dst, const int* idx, int B, int n, int JR, int JC, hipStream_t s);
void launch_scatter_results(const double* sub_out, const int* sub_level, int n, double* out, int* level, int B, const int* idx,
                            hipStream_t s);
// diagnostic: the compute units a stream's workgroups run on, seen[xcc * 256 + HW_ID bits 15:8] (api_kernels.hip)
void launch_cu_census(int* seen, int blocks, int spin, hipStream_t s);
// Filled by the per-width translation units (batch_w*.hip).
const BatchLaunchers* find_batch_launchers(int JR, int JC);
// prefix phase at the padded widths of the wide scan (16: 2 problems per wave, 32: one)
void launch_wide_prefix(const BatchParams& P, int width_padded, hipStream_t s);
// the wide prefix as a Kogge-Stone scan over composed elements (wide_prefix_scan.hip): few problems with many chunks
size_t wide_prefix_scan_workspace(int B, int nchunk, int width_padded);  // doubles; 0 = keep the sequential walk
int wide_prefix_scan_cap(int width_padded);                                // largest B x nchunk the scan is used for
int wide_prefix_scan_max_chunks(int width_padded);                         // most chunks one series is cut into
void launch_wide_prefix_scan(const BatchParams& P, int width_padded, hipStream_t s);
int wide_scan_max_width();
// fp32-state sequential sweep (a measurement for BASELINE config 5, wide_kernels.hip)
int wide_f32_probe_max_width();
void launch_wide_f32_probe(const BatchParams& P, int JR, int JC, double* out_logdet, double* out_quad, hipStream_t s);
void launch_wide_summarize(const BatchParams& P, int JR, int JC, hipStream_t s);
void launch_wide_correct(const BatchParams& P, int width_padded, hipStream_t s);
// widths 9..wide_max_width(): one wave per problem, sequential in n (wide_kernels.hip)
int wide_max_width();
void launch_wide_loglike(const BatchParams& P, int JR, int JC, hipStream_t s);

}  // namespace clr
