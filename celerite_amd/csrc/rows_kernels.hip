// celerite_amd/csrc/rows_kernels.hip -- CholeskySolver.compute at the widths the chunked scans do not take (general terms
// at any width, every width above 64; the plans' any-width route from a total width of 33): the reference's dynamic-width arm (cholesky.h:203), whose published
// benchmark goes to width 512 (examples/benchmark/run.py:37-39).  Sequential in n like the reference, parallel over the
// J^2 entries of the state, with S in REGISTERS:
//
//   layout     512 threads per workgroup; TPR adjacent lanes share a row of S, 32 columns each (64 registers); the padded
//              width is JP = 32 TPR, a workgroup holds 512 / TPR rows, G = TPR^2 / 16 workgroups hold the matrix:
//              width <= 128: TPR 4, one workgroup;  <= 256: TPR 8, 4 workgroups;  <= 512: TPR 16, 16;  <= 1024: TPR 32, 64.
//              S is kept whole (both triangles): twice the arithmetic of the triangle, but q = S u~ is then a row sum
//              with no mirrored contributions to exchange.
//   a step     (Y) S <- phi phi^T o (S + D_{n-1} W W^T) (cholesky.h:154-160) and q = S u~ (:163-175) in ONE pass over the
//              lane's 32 entries -- four vector instructions and 1.5 LDS reads (phi_k, W_k, u~_k: broadcast within a
//              column block, the blocks 34 doubles apart = conflict-free) per entry; the row sum over the TPR lanes by
//              shuffles; (X) D_n = d_n - u~ . q, W_n = (v~ - q) / D_n (:170-178), thread per row, and the NEXT sample's
//              phi, u~, v~: its exp and sincos calls one per thread, whole waves of one kind, issued before the barrier
//              (hidden behind the wait for the other workgroups), composed into the rows' features after it.
//              Two workgroup barriers per step; generic_kernels.hip's LDS-resident kernel needs five and leaves 3 of 4
//              threads idle in the matrix-vector product (width 128: 20.5 us per step there, the CPU 6.2).
//   G > 1      the workgroups exchange their rows of q and their share of u~ . q through a double-buffered array in HBM
//              (agent-scope atomic stores / loads: the 8 XCDs' L2s are not coherent for plain accesses) and meet at ONE
//              counter barrier per step; every workgroup then forms D_n and all of W_n itself, in the same order -- bit-
//              identical across workgroups, so a failed pivot (cholesky.h:176) is seen by all of them at the same step.
//              At most 64 workgroups on 256 compute units: co-resident by construction; the spin is bounded all the same
//              (status 3 instead of a hung queue).
//
// The sums run in another order than the reference's loops: results agree to rounding, not bit for bit (tests: 1e-10).
#include <hip/hip_runtime.h>

#include <math.h>

#include "../../include/celerite_hip.h"
#include "clr_generic_kernels.h"
#include "clr_options.h"
#include "clr_wide.h"

namespace clr {

namespace {

constexpr int ROWS_THREADS = 512;
constexpr int ROWS_COLS = 32;  // columns of S per lane

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the wave's global STORES
// (s_waitcnt vmcnt(0)) -- W, phi, u~ of the step, a microsecond of write latency per barrier that nobody needs to see
__device__ __forceinline__ void rows_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void agent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double agent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// column k of the per-column arrays in LDS: the 32-column blocks start 34 doubles apart, so that the TPR different blocks
// a wave reads in one ds_read_b128 fall into different banks
__device__ __forceinline__ int col_slot(int k) { return k + ((k >> 5) << 1); }

struct RowsExchange {
  double* q;           // [2][JP]  rows of q = S u~ of the step (parity of n)
  double* part;        // [2][G]   the workgroups' shares of u~ . q
  unsigned int* bar;   // arrivals, counted up over the whole run (zeroed before the launch)
};

// the row sum over the TPR adjacent lanes of a row, delivered to all of them (DPP butterflies, clr_wide.h)
template <int TPR>
__device__ __forceinline__ double rows_row_sum(double v) {
  v = dpp_add<DPP_QUAD_XOR1>(v);
  v = dpp_add<DPP_QUAD_XOR2>(v);
  if (TPR >= 8) v = dpp_add<DPP_HALF_MIRROR>(v);
  if (TPR >= 16) v = dpp_add<DPP_MIRROR>(v);
  if (TPR >= 32) v = swap_add16(v);
  return v;
}

// BATCH (one workgroup per problem of a plan, widths 33 .. 128: factor_rows_batch_kernel): nothing of the factor is stored,
// the diagonal is formed from the plan's series (`diag_in`, `dsum` = ((. + sum a_real) + sum a_comp) + jitter in the
// reference's order, cholesky.h:98-99, `A_in` added last), and the results go to the plan's output arrays.
struct RowsBatchOut {
  const double* diag_in;
  const double* A_in;
  double sum_ar, sum_ac, jitter;
  double *out_ll, *out_logdet, *out_quad;
  int* out_status;
};

template <int TPR, bool FAST, bool BLOCKED, bool BATCH>
__device__ __forceinline__ void factor_rows_body(const GenericProblem& g, RowsExchange X, const double* __restrict__ y, double* phi, double* u,
                                                 double* W, double* D, int* status, double* log_det, const RowsBatchOut& BO) {
  constexpr int JP = ROWS_COLS * TPR, RB = ROWS_THREADS / TPR, G = JP / RB, SLOTS = 34 * TPR;
  constexpr int RPT = (JP + ROWS_THREADS - 1) / ROWS_THREADS;  // rows (tasks) per thread in the per-row phases
  // Who does what besides the state's step (Y), in whole waves: the rows' pivots-and-W phase (X) belongs to threads
  // [0, JP); the exp / sincos TASKS of the sample two steps ahead to threads [TOFF, ...); COMPOSING the next sample's
  // phi, u~, v~ from the staged tasks to threads [COFF, ...).  At width <= 128 these are three different pairs of waves:
  // tasks and composition run while the (X) waves are still in (Y) or wait at the barrier, so that (X) itself -- on the
  // step's critical path, two waves of eight -- is just D, W and the hand-over (profiles/r06u_rows_kernel_phases.txt).
  constexpr int TOFF = JP <= 128 ? 128 : (JP <= 256 ? 256 : 0), COFF = JP <= 128 ? 256 : 0;
  __shared__ __attribute__((aligned(16))) double sphi[2][SLOTS], su[2][SLOTS], sw[SLOTS];  // features by the sample's parity
  __shared__ double sv[2][JP], sq[G == 1 ? JP : 1], spart[ROWS_THREADS / 64], sshare[G];
  __shared__ double sdot[ROWS_THREADS / 64];  // (y given) the waves' shares of u~_n . f_n, the forward sweep of dot_solve carried along
  // staged tasks by the sample's parity: decay of real row i | of complex term i - J_real; sin, cos of term jj
  __shared__ double sdecay[2][JP], ssin[2][JP / 2], scos[2][JP / 2];
  __shared__ int sabort;
  // t, the diagonal and y of 2 x 64 samples: a step's values come from here (a global load at the top of a step would be
  // on its critical path); the next tile is fetched 64 steps ahead
  __shared__ double tile_t[2][64], tile_d[2][64], tile_y[2][64];
  const int J = g.J, N = g.N, tid = threadIdx.x, wg = BATCH ? 0 : blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int part = tid % TPR, row = wg * RB + tid / TPR;  // (row-per-lane-group layout) this lane's row of S and block of columns
  const bool first = !BATCH && wg == 0;  // (who stores the factor)
  auto diag_at = [&](int i) {  // the full diagonal of sample i as handed over / formed from the plan's series
    if (!BATCH) return D[i];
    double d = ((BO.diag_in[i] + BO.sum_ar) + BO.sum_ac) + BO.jitter;
    if (BO.A_in) d += BO.A_in[i];
    return d;
  };
  const int JR = g.J_real, JC = g.J_comp, Wc = JR + 2 * JC, ndecay = JR + JC;

  double S[ROWS_COLS];
#pragma unroll
  for (int c = 0; c < ROWS_COLS; ++c) S[c] = 0.0;
  if (tid == 0) sabort = 0;

  // this thread's tasks (rate of a decay / frequency of a phase) and the constants of the rows it composes
  double trate[RPT];
  int tkind[RPT], tidx[RPT];  // 0 none, 1 decay, 2 phase; the task's index
  double ra[RPT], rb[RPT];
  int rkind[RPT], rsrc[RPT], crow[RPT];  // 0 none / padding, 1 real, 2 complex even, 3 complex odd, 4 general; index of the decay / term / general row; the row
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int tk = tid - TOFF + i * ROWS_THREADS;
    tkind[i] = 0; trate[i] = 0.0; tidx[i] = tk;
    if (tk >= 0 && tk < ndecay) { tkind[i] = 1; trate[i] = tk < JR ? g.c_real[tk] : g.c_comp[tk - JR]; }
    else if (tk >= ndecay && tk < ndecay + JC) { tkind[i] = 2; trate[i] = g.d_comp[tk - ndecay]; }
    const int k = tid - COFF + i * ROWS_THREADS;
    rkind[i] = -1; rsrc[i] = 0; ra[i] = 0.0; rb[i] = 0.0; crow[i] = k;
    if (k >= 0 && k < JP) {
      rkind[i] = 0;
      if (k < JR) { rkind[i] = 1; rsrc[i] = k; ra[i] = g.a_real[k]; }
      else if (k < Wc) { const int jj = (k - JR) >> 1; rkind[i] = 2 + ((k - JR) & 1); rsrc[i] = jj; ra[i] = g.a_comp[jj]; rb[i] = g.b_comp[jj]; }
      else if (k < J) { rkind[i] = 4; rsrc[i] = k - Wc; }
    }
  }
  auto run_tasks = [&](double t, double dx, int sb) {  // this thread's share of a sample's exp / sincos, into staging buffer sb
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      if (tkind[i] == 1) sdecay[sb][tidx[i]] = exp(-trate[i] * dx);
      else if (tkind[i] == 2) {
        double sd, cd;
        sincos_phase<FAST>(trate[i] * t, &sd, &cd);
        ssin[sb][tidx[i] - ndecay] = sd;
        scos[sb][tidx[i] - ndecay] = cd;
      }
    }
  };
  // phi, u~, v~ of this thread's rows at sample n (cholesky.h:129-152) from staging buffer sb into feature buffer n & 1;
  // the factor's phi, u~ of the move n - 1 -> n
  auto compose = [&](int n, int sb) {
    const int fb = n & 1;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      if (rkind[i] < 0) continue;
      double ph = 1.0, uu = 0.0, vv = 0.0;
      if (rkind[i] == 1) { ph = sdecay[sb][rsrc[i]]; uu = ra[i]; vv = 1.0; }
      else if (rkind[i] == 2 || rkind[i] == 3) {
        const double sd = ssin[sb][rsrc[i]], cd = scos[sb][rsrc[i]];
        ph = sdecay[sb][JR + rsrc[i]];
        uu = rkind[i] == 3 ? (ra[i] * sd - rb[i] * cd) : (ra[i] * cd + rb[i] * sd);
        vv = rkind[i] == 3 ? sd : cd;
      } else if (rkind[i] == 4) {
        uu = g.U[(long)rsrc[i] * N + n];
        vv = g.V[(long)rsrc[i] * N + n];
      }
      const int k = crow[i];
      sphi[fb][col_slot(k)] = ph;
      su[fb][col_slot(k)] = uu;
      sv[fb][k] = vv;
      if (first && k < J && n >= 1) { phi[(long)J * (n - 1) + k] = ph; u[(long)J * (n - 1) + k] = uu; }
    }
  };

  if (tid < 128) {
    tile_t[tid >> 6][tid & 63] = tid < N ? g.t[tid] : 0.0;
    tile_d[tid >> 6][tid & 63] = tid < N ? diag_at(tid) : 0.0;
    tile_y[tid >> 6][tid & 63] = (y && tid < N) ? y[tid] : 0.0;
  }
  __syncthreads();
  auto tile_at = [](const double (*tile)[64], int i) { return tile[(i >> 6) & 1][i & 63]; };

  // sample 0: cholesky.h:100-117; the features of sample 1, the tasks of sample 2
  double Dprev = tile_at(tile_d, 0);
  LogProduct lp;
  lp.init();
  lp.mul(Dprev);
  // the forward sweep of dot_solve (cholesky.h:343-357) for the vector announced by clr_solver_hint_rhs: f row by row in
  // the threads of (X) (every workgroup: all rows), x_n = y_n - u~_n . f_n from the waves' shares
  double fr[RPT], xm1 = y ? tile_at(tile_y, 0) : 0.0, quad = xm1 * (xm1 / Dprev), gsum = 0.0;
#pragma unroll
  for (int i = 0; i < RPT; ++i) fr[i] = 0.0;
  {
    const double t0 = tile_at(tile_t, 0), t1 = N > 1 ? tile_at(tile_t, 1) : t0, t2 = N > 2 ? tile_at(tile_t, 2) : t1;
    run_tasks(t0, 0.0, 0);
    __syncthreads();
    compose(0, 0);
    __syncthreads();
    const double value = 1.0 / Dprev;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int k = tid + i * ROWS_THREADS;
      if (k < JP) {
        const double w = sv[0][k] * value;
        sw[col_slot(k)] = w;
        if (first && k < J) W[k] = w;
      }
    }
    run_tasks(t1, t1 - t0, 1);
    __syncthreads();
    if (N > 1) compose(1, 1);
    run_tasks(t2, t2 - t1, 0);
    __syncthreads();
    if (y && N > 1) {  // cholesky.h:343-352: f_1 = phi (0 + W_0 x_0), and the rows' terms of u~_1 . f_1
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int k = tid + i * ROWS_THREADS;
        if (k < JP) {
          fr[i] = sphi[1][col_slot(k)] * (sw[col_slot(k)] * xm1);
          gsum += su[1][col_slot(k)] * fr[i];
        }
      }
    }
    if (y) {
      const double ws = row_sum_all<1>(gsum);
      if (lane == 0) sdot[wave] = ws;
    }
  }
  __syncthreads();

  for (int n = 1; n < N; ++n) {
    const int cur = n & 1, nxt = cur ^ 1;
    const double dn = tile_at(tile_d, n);  // (the full diagonal as handed over: cholesky.h:98-99)
    const bool fetch = (n & 63) == 0 && n >= 64 && tid < 64;  // the tile after this one (its buffer held the previous tile)
    double ft = 0.0, fd = 0.0, fy = 0.0;
    if (fetch) {
      const int i = n + 64 + tid;
      if (i < N) { ft = g.t[i]; fd = diag_at(i); fy = y ? y[i] : 0.0; }
    }
    double xn = 0.0;
    if (y) {
      xn = tile_at(tile_y, n);
      double dot = 0.0;
#pragma unroll
      for (int w = 0; w < ROWS_THREADS / 64; ++w) dot += sdot[w];
      xn -= dot;  // cholesky.h:353
    }
    const bool more = n + 1 < N;
    // the next sample's features from the tasks staged one step ago; the tasks of the sample after it (nobody reads
    // either target during this step's (Y)).  FIRST: the dependent chains of exp / sincos then fill the issue slots the
    // SIMD's other wave leaves in its (Y); after (Y) they would run alone, every other wave waiting at the barrier.
    if (more) compose(n + 1, nxt);
    if (n + 2 < N) {
      const double ta = tile_at(tile_t, n + 1), tb = tile_at(tile_t, n + 2);
      run_tasks(tb, tb - ta, cur);
    }
    // ---- (Y) the state's step and q = S u~, one pass ----------------------------------------------------------------
    if constexpr (BLOCKED) {
      // width <= 128, one workgroup: a lane holds 4 rows x 8 columns -- 12 (columns) + 4 (rows) LDS quadwords per lane
      // and step where the row-per-lane-group layout below reads 48 + 2, at the price of four row sums over 16 lanes
      const int cb = tid & 15, r0 = (tid >> 4) * 4;
      const double2* pk2 = reinterpret_cast<const double2*>(&sphi[cur][col_slot(8 * cb)]);
      const double2* wk2 = reinterpret_cast<const double2*>(&sw[col_slot(8 * cb)]);
      const double2* uk2 = reinterpret_cast<const double2*>(&su[cur][col_slot(8 * cb)]);
      const double2* pr2 = reinterpret_cast<const double2*>(&sphi[cur][col_slot(r0)]);
      const double2* wr2 = reinterpret_cast<const double2*>(&sw[col_slot(r0)]);
      double pk[8], wk[8], uk[8], prr[4], dwr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 a = pk2[c], bb = wk2[c], cc = uk2[c];
        pk[2 * c] = a.x; pk[2 * c + 1] = a.y; wk[2 * c] = bb.x; wk[2 * c + 1] = bb.y; uk[2 * c] = cc.x; uk[2 * c + 1] = cc.y;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double2 a = pr2[c], bb = wr2[c];
        prr[2 * c] = a.x; prr[2 * c + 1] = a.y; dwr[2 * c] = Dprev * bb.x; dwr[2 * c + 1] = Dprev * bb.y;
      }
      double qr[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          const double s0 = prr[rr] * (pk[c] * fma(dwr[rr], wk[c], S[8 * rr + c]));
          const double s1 = prr[rr] * (pk[c + 1] * fma(dwr[rr], wk[c + 1], S[8 * rr + c + 1]));
          S[8 * rr + c] = s0;
          S[8 * rr + c + 1] = s1;
          a0 = fma(s0, uk[c], a0);
          a1 = fma(s1, uk[c + 1], a1);
        }
        qr[rr] = rows_row_sum<16>(a0 + a1);
      }
      double mine = 0.0;
      if (cb == 0) {
        const double2* ur2 = reinterpret_cast<const double2*>(&su[cur][col_slot(r0)]);
        const double2 u01 = ur2[0], u23 = ur2[1];
        mine = (u01.x * qr[0] + u01.y * qr[1]) + (u23.x * qr[2] + u23.y * qr[3]);  // (padding rows: u~ = 0)
        sq[r0] = qr[0]; sq[r0 + 1] = qr[1]; sq[r0 + 2] = qr[2]; sq[r0 + 3] = qr[3];
      }
      const double wsum = row_sum_all<1>(mine);
      if (lane == 0) spart[wave] = wsum;
    } else {
      const double pr = sphi[cur][col_slot(row)];
      const double dw = Dprev * sw[col_slot(row)];
      double acc0 = 0.0, acc1 = 0.0;
      {
        const double2* pk2 = reinterpret_cast<const double2*>(&sphi[cur][34 * part]);
        const double2* wk2 = reinterpret_cast<const double2*>(&sw[34 * part]);
        const double2* uk2 = reinterpret_cast<const double2*>(&su[cur][34 * part]);
#pragma unroll
        for (int c = 0; c < ROWS_COLS / 2; ++c) {
          const double2 pk = pk2[c], wk = wk2[c], uk = uk2[c];
          const double s0 = pr * (pk.x * fma(dw, wk.x, S[2 * c]));
          const double s1 = pr * (pk.y * fma(dw, wk.y, S[2 * c + 1]));
          S[2 * c] = s0;
          S[2 * c + 1] = s1;
          acc0 = fma(s0, uk.x, acc0);
          acc1 = fma(s1, uk.y, acc1);
        }
      }
      const double q = rows_row_sum<TPR>(acc0 + acc1);
      const double mine = (part == 0) ? su[cur][col_slot(row)] * q : 0.0;  // (padding rows: u~ = 0)
      const double wsum = row_sum_all<1>(mine);
      if (lane == 0) spart[wave] = wsum;
      if (part == 0) {
        if (G == 1) sq[row] = q;
        else agent_store(X.q + (size_t)(n & 1) * JP + row, q);
      }
    }
    if (G == 1) rows_lds_barrier();
    else __syncthreads();  // (the rows of q this wave stored must have left before thread 0 announces the workgroup's arrival)
    double total = 0.0;
    if (G == 1) {
#pragma unroll
      for (int w = 0; w < ROWS_THREADS / 64; ++w) total += spart[w];
    } else {
      if (tid == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < ROWS_THREADS / 64; ++w) s += spart[w];
        agent_store(X.part + (size_t)(n & 1) * G + wg, s);
        __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope by default on this target: the rows of q the other waves stored, too)
        __hip_atomic_fetch_add(X.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int want = (unsigned int)G * (unsigned int)n;
        long spins = 0;
        while (__hip_atomic_load(X.bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++spins > (1L << 26)) { status[0] = 3; sabort = 1; break; }  // (a workgroup that never arrives: give up rather than hang the queue)
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
      }
      if (wave == 0 && lane < G) sshare[lane] = agent_load(X.part + (size_t)(n & 1) * G + lane);
      rows_lds_barrier();
#pragma unroll 8
      for (int w = 0; w < G; ++w) total += sshare[w];  // (every thread, every workgroup: the same order)
    }
    // ---- (X) the pivot and W_n ------------------------------------------------------------------------------------------
    const double Dn = dn - total;
    if (Dn < 0.0 || (G > 1 && sabort)) {  // cholesky.h:176
      if (first && tid == 0) { if (!sabort) status[0] = 1; log_det[0] = NAN; }
      if (BATCH && tid == 0) { BO.out_status[0] = 2; BO.out_ll[0] = -INFINITY; BO.out_logdet[0] = NAN; BO.out_quad[0] = NAN; }
      return;
    }
    gsum = 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int k = tid + i * ROWS_THREADS;
      if (k < JP) {
        const double qk = (G == 1) ? sq[k] : agent_load(X.q + (size_t)(n & 1) * JP + k);
        const double w = (sv[cur][k] - qk) / Dn;  // cholesky.h:170-178
        sw[col_slot(k)] = w;
        if (first && k < J) W[(long)J * n + k] = w;
        if (y && more) {
          fr[i] = sphi[nxt][col_slot(k)] * (fr[i] + w * xn);  // cholesky.h:350-352
          gsum += su[nxt][col_slot(k)] * fr[i];
        }
      }
    }
    if (y) {
      quad += xn * xn / Dn;  // cholesky.h:355
      const double ws = row_sum_all<1>(gsum);
      if (lane == 0) sdot[wave] = ws;  // (read at the top of the next step, behind the barrier below)
    }
    if (first && tid == 0) D[n] = Dn;
    if (tid == 0) lp.mul(Dn);
    if (fetch) {
      const int bsel = ((n >> 6) + 1) & 1;
      tile_t[bsel][tid] = ft; tile_d[bsel][tid] = fd; tile_y[bsel][tid] = fy;
    }
    Dprev = Dn;
    rows_lds_barrier();
  }
  if (first && tid == 0) { status[0] = 0; log_det[0] = lp.log_value(); log_det[1] = quad; }
  if (BATCH && tid == 0) {
    const double ld = lp.log_value();
    double ll = -0.5 * (quad + ld + N * 1.8378770664093453);
    if (!isfinite(ld) || !isfinite(ll)) ll = -INFINITY;  // celerite.py:211-218
    BO.out_status[0] = 0; BO.out_logdet[0] = ld; BO.out_quad[0] = quad; BO.out_ll[0] = ll;
  }
}

template <int TPR, bool FAST, bool BLOCKED = false>
__global__ void __launch_bounds__(ROWS_THREADS) factor_rows_kernel(GenericProblem g, RowsExchange X, const double* __restrict__ y, double* phi, double* u,
                                                                   double* W, double* D, int* status, double* log_det) {
  factor_rows_body<TPR, FAST, BLOCKED, false>(g, X, y, phi, u, W, D, status, log_det, RowsBatchOut{});
}

// one workgroup per problem of a plan (total width 33 .. 128, general terms included): the fused log-likelihood
template <bool FAST>
__global__ void __launch_bounds__(ROWS_THREADS) factor_rows_batch_kernel(const GenericBatch G) {
  const int b = blockIdx.x;
  GenericProblem g;
  g.N = G.N;
  g.J_real = G.J_real; g.J_comp = G.J_comp; g.J_general = G.J_general;
  g.J = G.J_real + 2 * G.J_comp + G.J_general;
  g.a_real = G.a_real + (long)b * G.J_real; g.c_real = G.c_real + (long)b * G.J_real;
  g.a_comp = G.a_comp + (long)b * G.J_comp; g.b_comp = G.b_comp + (long)b * G.J_comp;
  g.c_comp = G.c_comp + (long)b * G.J_comp; g.d_comp = G.d_comp + (long)b * G.J_comp;
  g.U = G.U ? G.U + (long)b * G.U_stride : nullptr;
  g.V = G.V ? G.V + (long)b * G.V_stride : nullptr;
  g.t = G.t + (long)b * G.t_stride;
  RowsBatchOut BO;
  BO.diag_in = G.diag + (long)b * G.diag_stride;
  BO.A_in = G.A ? G.A + (long)b * G.A_stride : nullptr;
  double sum_ar = 0.0, sum_ac = 0.0;  // cholesky.h:98: the reference's summation order
  for (int j = 0; j < G.J_real; ++j) sum_ar += g.a_real[j];
  for (int j = 0; j < G.J_comp; ++j) sum_ac += g.a_comp[j];
  BO.sum_ar = sum_ar; BO.sum_ac = sum_ac; BO.jitter = G.jitter[b];
  BO.out_ll = G.out_ll + b; BO.out_logdet = G.out_logdet + b; BO.out_quad = G.out_quad + b; BO.out_status = G.out_status + b;
  factor_rows_body<4, FAST, true, true>(g, RowsExchange{nullptr, nullptr, nullptr}, G.y + (long)b * G.y_stride, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, BO);
}

int rows_tpr(int J) { return J <= 128 ? 4 : (J <= 256 ? 8 : (J <= 512 ? 16 : 32)); }

}  // namespace

bool factor_rows_supported(int J) { return J >= 1 && J <= 1024; }

// doubles of workspace (q | shares | the counter in the last slot)
size_t factor_rows_workspace_doubles(int J) {
  const int tpr = rows_tpr(J), JP = ROWS_COLS * tpr, G = tpr * tpr / 16;
  return (size_t)2 * JP + 2 * G + 2;
}

// D arrives initialised to the full diagonal (cholesky.h:98-99); status[0] = 1: a pivot D_n < 0 (n >= 1), 3: the
// workgroups lost each other (never seen; reported as a HIP error by the caller).  log_det[0] = sum log D_n; with y given
// (device, [N]) log_det[1] = y^T K^-1 y: the forward sweep of dot_solve (cholesky.h:343-357) carried along.
void launch_factor_rows(const GenericProblem& g, int fast_trig, const double* y, double* workspace, double* phi, double* u, double* W, double* D,
                        int* status, double* log_det, hipStream_t s) {
  const int tpr = rows_tpr(g.J), JP = ROWS_COLS * tpr, G = tpr * tpr / 16;
  RowsExchange X;
  X.q = workspace;
  X.part = workspace + (size_t)2 * JP;
  X.bar = reinterpret_cast<unsigned int*>(workspace + (size_t)2 * JP + 2 * G);
  (void)hipMemsetAsync(X.bar, 0, 2 * sizeof(double), s);
  (void)hipMemsetAsync(status, 0, sizeof(int), s);
#define CLR_ROWS_LAUNCH(T)                                                                                                                  \
  do {                                                                                                                                      \
    if (fast_trig) hipLaunchKernelGGL((factor_rows_kernel<T, true>), dim3(G), dim3(ROWS_THREADS), 0, s, g, X, y, phi, u, W, D, status, log_det); \
    else hipLaunchKernelGGL((factor_rows_kernel<T, false>), dim3(G), dim3(ROWS_THREADS), 0, s, g, X, y, phi, u, W, D, status, log_det);          \
  } while (0)
  if (tpr == 4) {  // (one workgroup: the 4 x 8 register blocks; CLR_ROWS_NO_BLOCKS=1 keeps the row-per-lane-group layout for A/B)
    if (clr::option("CLR_ROWS_NO_BLOCKS")) CLR_ROWS_LAUNCH(4);
    else if (fast_trig) hipLaunchKernelGGL((factor_rows_kernel<4, true, true>), dim3(G), dim3(ROWS_THREADS), 0, s, g, X, y, phi, u, W, D, status, log_det);
    else hipLaunchKernelGGL((factor_rows_kernel<4, false, true>), dim3(G), dim3(ROWS_THREADS), 0, s, g, X, y, phi, u, W, D, status, log_det);
  }
  else if (tpr == 8) CLR_ROWS_LAUNCH(8);
  else if (tpr == 16) CLR_ROWS_LAUNCH(16);
  else CLR_ROWS_LAUNCH(32);
#undef CLR_ROWS_LAUNCH
}

// the plans' fused log-likelihood at total widths 33 .. 128 (general terms included): one workgroup per problem
bool factor_rows_batch_supported(int J_total) { return J_total >= 33 && J_total <= 128; }
void launch_factor_rows_batch(const GenericBatch& G, int fast_trig, hipStream_t s) {
  if (fast_trig) hipLaunchKernelGGL((factor_rows_batch_kernel<true>), dim3(G.B), dim3(ROWS_THREADS), 0, s, G);
  else hipLaunchKernelGGL((factor_rows_batch_kernel<false>), dim3(G.B), dim3(ROWS_THREADS), 0, s, G);
}

}  // namespace clr
