// celerite_amd/csrc/wide_kernels.hip -- batched fused log-likelihood for widths 9..64
// (BASELINE config 5: 16 complex terms = width 32, N = 1e5, 256 problems).
//
// At these widths one problem's state S (cholesky.h:154-160) no longer fits a lane, so
// the mapping turns around: ONE WAVE PER PROBLEM, sequential in n exactly as the
// reference (cholesky.h:126-179 fused with dot_solve, :348-357), with the W x W matrix
// distributed over the 64 lanes in registers:
//     WMAX = 16: 4 lanes per row, 4 columns each      (LPR = 4, COLS = 4)
//     WMAX = 32: 2 lanes per row, 16 columns each     (LPR = 2, COLS = 16)
//     WMAX = 64: 1 lane  per row, 64 columns          (LPR = 1, COLS = 64)
// S is kept in FULL (not packed-symmetric) storage: every lane updates its own row
// segment, no lane idles on a triangle.
//
// One step (state before sample n: S = P_n, f = f_n, the reference's S and f after
// their updates at step n; features u~, v~ at t_n, decay phi for t_n -> t_{n+1}):
//     q = S u ; D = a - u.q ; z = v - q ; w = z / D ; x = y_n - u.f
//     S <- Phi (S + z w^T) Phi ; f <- Phi (f + w x)
// * a sample's features (sin/cos of the absolute phase, cholesky.h:137; exp of the step,
//   :130,140) are evaluated ONCE per row -- the row's lanes split the next samples between
//   them, once per LPR steps -- or once per complex term when there are no real terms
//   (round 5: wide_scan_body, FB / FBN / PAIRED; every lane did all of it on every step before);
// * the three vectors a lane needs for its COLUMNS (u, phi, phi*w) cross the wave
//   through small LDS buffers (one write + broadcast b128 reads; a wave's LDS
//   operations execute in program order, so no barrier is needed -- but the COMPILER may
//   reorder a load against another lane's store across divergent arms: see the wave
//   barrier in the renormalisation); u and phi do not depend on the state and are
//   published a batch ahead, so phi*w is the only exchange on a step's critical path;
// * row dot products finish with log2(LPR) DPP butterfly stages, u.q and u.f with
//   the remaining DPP stages inside 16 lanes and four v_readlane pairs across them
//   (a lone wave would wait ~100 cycles on every ds_bpermute of a __shfl_xor);
// * t, diag, y are fetched 64 samples at a time (one coalesced 512-B load per array)
//   and handed out through LDS (a time ring for the per-lane samples of the feature
//   batch, K(0) + diag and y by two uniform reads per step; v_readlane before round 5).
// Flops per step ~ 3.5 W^2; vector instructions per step at W = 32: ~130 (lazy summarize
// with riders, profiles/r05k_wide_isa_mix.txt; round 4: 190, round 1: ~240).
// B problems alone use B of the chip's 1024 SIMDs; for smaller batches (widths <= 32) the
// time axis is cut into chunks as in the narrow scan: MODE 1 of the kernel also builds the
// chunk's transfer element and zero-start sums, prefix_coop_kernel<16 | 32> chains the chunks,
// wide_correct_kernel turns the sums into the true contributions, and MODE 0 replays only the
// problems it could not certify (profiles/r01s_wide_scan.log: config 5, 80.6 -> 21.5 ms).
#include "../../include/celerite_hip.h"
#include "clr_options.h"
#include "clr_batch_kernels.h"
#include "clr_wide.h"

// Round-4 build switches (A/B builds: -DCLR_WIDE_JM_MFMA=0 etc.; profiles/r04b_wide_ab.txt)
#ifndef CLR_WIDE_JM_MFMA
#define CLR_WIDE_JM_MFMA 1   // width 32 lazy summarize: Jm -= R D^-1 R^T as rank-16 updates on v_mfma_f64_16x16x4
#endif
#ifndef CLR_WIDE_PACKED_SUMS
#define CLR_WIDE_PACKED_SUMS 1  // u.q and u.f reduced together (even / odd lanes), one DPP tree instead of two
#endif
#ifndef CLR_WIDE_LOGPROD_WINDOW
#define CLR_WIDE_LOGPROD_WINDOW 1  // lazy summarize: plain product of a block's 16 pivots, one frexp per block
#endif
// Round 5: the lazy summarize's per-row features (rotation of the (cos, sin) pair, the accumulated decay, u, v) are
// evaluated once per LPR steps -- the LPR lanes of a row each take ONE of the next LPR samples -- instead of by every
// lane on every step (wide_scan_body, FB)
#ifndef CLR_WIDE_FEATURE_BATCH
#define CLR_WIDE_FEATURE_BATCH 1
#endif

namespace clr {

namespace {

typedef double mfma_acc_t __attribute__((ext_vector_type(4)));

// One row's features: u~, v~ (cholesky.h:129-147) at t and the decay to t + dx, without
// selects: u = u0 + uc cos(d t) + us sin(d t), v = v0 + vc cos + vs sin with per-row
// constants (real row: u0 = a, v0 = 1; cos row: uc = a, us = b, vc = 1; sin row:
// uc = -b, us = a, vs = 1; padding rows: all 0, c = 0 so phi = 1 and S stays 0).
struct RowCoeffs {
  double u0, uc, us, v0, vc, vs, c, d;
};
template <bool FAST>
__device__ __forceinline__ void row_features(const RowCoeffs& r, double t, double dx, double* u,
                                             double* v, double* phi) {
  double sd, cs;
  sincos_phase<FAST>(r.d * t, &sd, &cs);
  const double x = -r.c * dx;
  // densely sampled series: the 6-FMA polynomial (see features_phi_distinct); wave-uniform choice
  *phi = CLR_WAVE_ALL(fabs(x) < 0.0078125) ? exp_small(x) : exp(x);
  *u = fma(r.uc, cs, fma(r.us, sd, r.u0));
  *v = fma(r.vc, cs, fma(r.vs, sd, r.v0));
}

// MODE 0: the log-likelihood recurrence over samples [n_lo, n_hi) of chunk blockIdx.x from a
//         given start state (zero for the first chunk).  With one chunk the kernel writes the
//         problem's results itself; with several it writes the chunk's partial sums for
//         finalize_kernel ("replay" of the scan).
// MODE 1: "summarize": the same recurrence from the ZERO state plus the chunk's transfer
//         element (A, b, C, eta, Jm) in the layout of the narrow scan kernels at width
//         J = WMAX (padding rows behave as identity), so that prefix_coop_kernel<WMAX> can
//         chain the chunks.  A is held transposed (lane (j, seg) owns A[seg*COLS ..][j]):
//         r_j = u . A[:, j] is then a lane-local dot product like q, and the update
//         A[i][j] <- phi_i A[i][j] - (phi w)_i r_j reuses the phi / phi*w reads of the S update.
// LAZY (summarize on densely sampled series only; host-checked max c dx < 2^-7, max d dx < 2^-5): the decay is
// factored out of the state exactly as in clr_split_kernels.h -- S = Psi Sbar Psi, A = Psi Abar, f = Psi fbar with Psi
// the decay accumulated since the last renormalisation (every 16 steps) -- so the S / A updates are one FMA per
// entry instead of two MULs + FMA and phi is neither published nor read back; and every row's (cos, sin) pair advances
// by a small-angle rotation, re-anchored with the full sincos every 16 samples.
template <bool LAZY>
__device__ __forceinline__ void decay_pair(double x, double* phi, double* phinv) {
  if (CLR_WAVE_ALL(fabs(x) < 0.0078125)) {
    const double x2 = x * x;
    const double ch = fma(x2, fma(x2, fma(x2, 1.0 / 720.0, 1.0 / 24.0), 0.5), 1.0);
    const double sh = x * fma(x2, fma(x2, 1.0 / 120.0, 1.0 / 6.0), 1.0);
    *phi = ch + sh;
    *phinv = ch - sh;
  } else {
    *phi = exp(x);
    *phinv = exp(-x);
  }
}

// d = a * b + c as ONE v_fma_f64 with all operands in registers.  The compiler turns fma(x, acc, K) with a
// loop-invariant K into v_mov_b64 tmp, K ; v_fmac_f64 tmp, x, acc -- two issue slots for one; 13 such copies per
// feature batch in the lazy summarize (profiles/r05k_wide_isa_mix.txt).
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max(a, |b|) in one instruction (fmax(a, fabs(b)) costs a canonicalising v_max_f64 a, a, a on top)
__device__ __forceinline__ double max_abs(double a, double b) {
  double d;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// a value the compiler can no longer see through: a double constant kept as ONE register pair (it otherwise shares
// equal 32-bit halves between constants and patches them together with v_mov_b32 inside the loop), a loop counter it
// must keep scalar
__device__ __forceinline__ double opaque_v(double x) { asm("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm("" : "+s"(x)); return x; }
// the Taylor coefficients of the FB polynomials, materialised once per kernel
struct FbConsts {
  double k6, k24, k120, k720, k5040, k40320, k362880, k3628800;
  __device__ __forceinline__ void init(bool wide) {
    k6 = opaque_v(1.0 / 6.0); k24 = opaque_v(1.0 / 24.0); k120 = opaque_v(1.0 / 120.0); k720 = opaque_v(1.0 / 720.0);
    k5040 = opaque_v(1.0 / 5040.0); k40320 = opaque_v(1.0 / 40320.0);
    k362880 = wide ? opaque_v(1.0 / 362880.0) : 0.0; k3628800 = wide ? opaque_v(1.0 / 3628800.0) : 0.0;
  }
};
__device__ __forceinline__ double fma3n(double a, double b, double c) {  // a * b - c
  double d;
  asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double fnma3(double a, double b, double c) {  // c - a * b
  double d;
  asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ double min3(double a, double b) {  // min(a, b), no canonicalising copies (both operands are results of arithmetic)
  double d;
  asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ double fnma3_one(double a, double b) {  // 1 - a * b (inline constant)
  double d;
  asm("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ double fnma3_half(double a, double b) {  // 1/2 - a * b
  double d;
  asm("v_fma_f64 %0, -%1, %2, 0.5" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// FB: the decay over a lane's own LPR-step interval, |x| < NB * 2^-7 (host-checked per step: max c dx < 2^-7), so no
// range test and no exp(): cosh / sinh series, truncation below 5e-17 (NB = 2: next terms x^7 / 5040, x^8 / 40320 at
// |x| <= 2^-6; NB = 4 carries one more term each for |x| <= 2^-5)
template <int NB>
__device__ __forceinline__ void decay_pair_nb(const FbConsts& K, double x, double* phi, double* phinv) {
  const double x2 = x * x;
  double ch, sh;
  if (NB <= 2) {
    ch = fma(x2, fma(x2, fma3(x2, K.k720, K.k24), 0.5), 1.0);
    sh = x * fma(x2, fma3(x2, K.k120, K.k6), 1.0);
  } else {
    ch = fma(x2, fma(x2, fma3(x2, fma3(x2, K.k40320, K.k720), K.k24), 0.5), 1.0);
    sh = x * fma(x2, fma3(x2, fma3(x2, K.k5040, K.k120), K.k6), 1.0);
  }
  *phi = ch + sh;
  *phinv = ch - sh;
}
// FB: (cos, sin) of a lane's own rotation step, |a| < NB * 2^-5 (host-checked per step: max d dx < 2^-5): absolute
// truncation below 5e-17 (NB = 2: a^9 / 9!, a^10 / 10! at |a| <= 2^-4; NB = 4 one more term each for |a| <= 2^-3)
template <int NB>
__device__ __forceinline__ void small_sincos_nb(const FbConsts& K, double a, double* sn, double* cn) {
  const double a2 = a * a;
  // sin a / a = 1 - a2 (1/6 - a2 (1/120 - a2 (1/5040 - a2 / 362880))) ; cos a = 1 - a2 (1/2 - a2 (1/24 - a2 (1/720 - a2 (1/40320 - a2 / 3628800))))
  // (the same coefficients and Horner order as the signed form; c - a b is the instruction's own negate modifier)
  if (NB <= 2) {
    *sn = a * fnma3_one(a2, fnma3(a2, fnma3(a2, K.k5040, K.k120), K.k6));
    *cn = fnma3_one(a2, fnma3_half(a2, fnma3(a2, fnma3(a2, K.k40320, K.k720), K.k24)));
  } else {
    *sn = a * fnma3_one(a2, fnma3(a2, fnma3(a2, fnma3(a2, K.k362880, K.k5040), K.k120), K.k6));
    *cn = fnma3_one(a2, fnma3_half(a2, fnma3(a2, fnma3(a2, fnma3(a2, K.k3628800, K.k40320), K.k720), K.k24)));
  }
}

// first sample of chunk c (c == nchunk: N).  Uniform chunks of L samples, or (L0 > 0) a first chunk of L0 samples
// followed by chunks of L
__device__ __forceinline__ int wide_chunk_begin(const BatchParams& P, int c) {
  const long n = (P.L0 > 0 && c > 0) ? (long)P.L0 + (long)(c - 1) * P.L : (long)c * P.L;
  return n < P.N ? (int)n : P.N;
}

// RIDERS == false (summarize only): the chunk that starts at sample 0.  Its start state IS the zero state, so the
// prefix takes (C, b) of its element as they are and nothing ever reads A, Jm, eta: the riders -- half of a
// summarize step's state FMAs -- are not carried, and the host makes that chunk longer in return (BatchParams::L0).
// GEN: general semiseparable terms (cholesky.h:65-72,114-116,148-152): rows W_c .. W_c + J_general - 1 behind the
// celerite rows carry phi = 1 and per-sample features u = U[j][n], v = V[j][n]; A[n] joins the diagonal.  A general
// row is a "real" row (d = 0: cos = 1, sin = 0, c = 0: phi = 1) whose constants u0, v0 are REPLACED every step by
// the row's next sample, fetched GEN_PF steps ahead through a register queue -- nothing else in the step changes.
constexpr int GEN_PF = 6;
// NW (round 5): waves per (problem, chunk).  NW = 2 at WMAX = 64: a workgroup of 128 lanes, TWO lanes per row with 32
// columns each -- S and A^T are 2 x 32 doubles per lane (128 registers) instead of 2 x 64 (all 256 architectural
// registers, 329 v_accvgpr moves per step), two waves per SIMD fit, and both the summarize and the replay flavour run
// without scratch.  The price is two workgroup barriers per step: the row sums u.q, u.f cross the two waves through LDS
// (xbuf; both waves add the two partial sums in the same order, so D is bit-identical in both), and the step's w (and r)
// must be complete before the rank-1 updates read them.  u, phi are published a step ahead as before (their writes and
// the previous step's reads are separated by those barriers).
// The body's LDS arrays, ONE set per kernel: a kernel runs the body without riders on its first chunk and the body with
// riders on the others, a workgroup only ever one of them -- as static arrays of the body they were allocated twice
// (21.5 KB with the PAIRED slots: seven workgroups per CU instead of the eight that give every SIMD its two waves).
template <int WMAX, bool LAZY, bool FBF, int NSLOT, bool JMMF, bool RBUF, int NW>
struct WideLds {
  alignas(16) double fuvb[LAZY ? 2 : 3][FBF ? NSLOT : 1][FBF ? WMAX : 2];  // (FB) [0]: ubar | u, [1]: vbar | v, [2]: (plain flavour) phi (one address register serves all)
  alignas(16) double rblk[JMMF ? 16 * WMAX : 2];
  alignas(16) double rsblk[JMMF ? 16 * WMAX : 2];
  alignas(16) double ubuf[2][FBF ? 2 : WMAX];
  alignas(16) double pbuf[2][FBF ? 2 : WMAX];
  alignas(16) double wbuf[WMAX];
  alignas(16) double rbuf[RBUF ? WMAX : 2];
  alignas(16) double psibuf[LAZY ? WMAX : 2];
  double tring[FBF ? 128 : 2];
  double dtile[FBF ? 64 : 2], ytile[FBF ? 64 : 2];  // (FB) the tile's K(0) + diag and y: a step's pair by two LDS reads (v_readlane: four vector slots)
  double xbuf[NW == 2 ? 8 : 1];  // (NW = 2) the two waves' partial row sums: [wave][u.q | u.f] (and, at the end, their residual maxima)
};
template <class L>
__device__ __forceinline__ L& wide_lds() {
  __shared__ L lds;
  return lds;
}

template <int WMAX, bool FAST, int MODE, bool LAZY, bool RIDERS, bool GEN, int NW = 1, bool PAIRED = false, bool GAPS = false>
__device__ __forceinline__ void wide_scan_body(const BatchParams& P, int JR, int JC) {
  static_assert(!LAZY || MODE == 1, "the lazy decay is a summarize flavour");
  static_assert(NW == 1 || (NW == 2 && WMAX == 64), "two waves per (problem, chunk): the padded width 64");
  constexpr bool RID = MODE == 1 && RIDERS;
  constexpr int LPR = 64 * NW / WMAX, COLS = WMAX / LPR;
  constexpr int J = WMAX, SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  // Jm on the matrix cores (width 32, lazy summarize): the r of a 16-step block and -r / D are parked in LDS
  // [step][row]; at the block's end Jm += R (-D^-1 R)^T is 3 output tiles ((0,0), (0,1), (1,1): Jm is symmetric) x 4
  // k-steps = 12 v_mfma_f64_16x16x4, operands read from LDS in the instruction's lane layout.  Replaces 16 FMAs and
  // eight b128 LDS reads per lane and STEP.  (A and r stay on the VALU: r_n needs the up-to-date A, and a blocked
  // WY form costs R0 = A0^T U, G = W^T U and a forward substitution on top -- 40 MFMA + 120 FMA per block against
  // 32 FMA per step, with fp64 MFMA only 1.28x the VALU's FMA rate and not overlapping it: profiles/r04a_mfma_overlap.txt.)
  constexpr bool JMM = RID && LAZY && (WMAX == 32 || WMAX == 64) && CLR_WIDE_JM_MFMA;
  constexpr int NTL = WMAX / 16, NTILE = NTL * (NTL + 1) / 2;  // (JMM) 16 x 16 tiles per side / of the upper triangle of Jm
  constexpr int NTW = NTILE / NW;                              // ... tiles per wave (NW = 2: every other tile)
  constexpr bool PACKED = LPR >= 2 && CLR_WIDE_PACKED_SUMS;
  constexpr bool LPWIN = LAZY && CLR_WIDE_LOGPROD_WINDOW;
#ifndef CLR_WIDE_RENORM_STEPS
#define CLR_WIDE_RENORM_STEPS 64
#endif
  // (LAZY) steps between renormalisations: a multiple of 16.  The padded width 64 keeps 16 and the one block it had:
  // its kernels sit at 256 registers, and the second block costs them 500 more spilled registers (98 ms instead of 41)
  constexpr int RN = WMAX == 64 ? 16 : CLR_WIDE_RENORM_STEPS;
  // FB (lazy summarize, celerite rows only, two or four lanes per row): a step's features are functions of the times
  // alone, and every lane of a row used to evaluate all of them on every step -- ~60 of the step's 190 vector
  // instructions.  Now the row's LPR lanes split the NEXT LPR samples between them: once per LPR steps lane `seg`
  // advances ITS OWN (cos, sin) pair by d (t_m - t_(m - LPR)) and ITS OWN accumulated decay Psi by
  // exp(-c (t_m - t_(m - LPR))) for sample m = (first of the batch) + seg, and publishes ubar_m = Psi u_m and
  // vbar_m = v_m / Psi in LDS slots [m mod 2 LPR][row]; a step reads its own row's pair and the row of ubar back.  No
  // per-step decay phi is formed at all (Psi at the renormalisation is the lane's Psi carried to t_(n + 1)); the times
  // come from a 128-entry ring in LDS because every lane needs a different sample's.
  // The PLAIN flavour (sparse series' summarize, every replay / sequential sweep: FBN) splits the same way -- there a
  // sample's features are a full sincos of the absolute phase and an exp of the step (cholesky.h:130,137,140), ~100 of
  // the step's vector instructions, and nothing is carried from sample to sample: u, v, phi go to the slots as they are.
  // General rows ride along: their "features" are the row's U, V samples (no arithmetic), fetched a batch ahead.
  constexpr bool FBA = LPR >= 2 && CLR_WIDE_FEATURE_BATCH;
  constexpr bool FB = FBA && LAZY, FBN = FBA && !LAZY;
  // PAIRED (host: no real terms, two lanes per row): the cos and the sin row of a complex term share c and d, hence the
  // (cos, sin) pair and Psi -- the term's FOUR lanes split the next four samples, and each publishes BOTH rows' entries
  // of its sample: u = a cos + b sin | a sin - b cos, v = cos | sin (cholesky.h:143-146)
  constexpr bool PR = FBA && PAIRED && LPR == 2;
  constexpr int NB = FBA ? (PR ? 2 * LPR : LPR) : 1, NSLOT = 2 * NB;
  // u and phi of a step are written one step AHEAD (they do not depend on the state),
  // double-buffered; phi * w is the one true exchange of a step.  A wave's LDS
  // operations execute in program order, so no barrier or explicit wait is needed.
  // (sizes: those of the body WITH riders, the superset)
  constexpr bool JMM_ANY = MODE == 1 && LAZY && (WMAX == 32 || WMAX == 64) && CLR_WIDE_JM_MFMA;
  auto& lds_ = wide_lds<WideLds<WMAX, LAZY, FBA, NSLOT, JMM_ANY, MODE == 1 && !JMM_ANY, NW>>();
  auto& fub = lds_.fuvb[0];
  auto& fvb = lds_.fuvb[1];
  auto& fpb = lds_.fuvb[LAZY ? 1 : 2];  // (the plain flavour's plane; never touched by the lazy one)
  auto& tring = lds_.tring;
  auto& dtile = lds_.dtile;
  auto& ytile = lds_.ytile;
  auto& rblk = lds_.rblk;
  auto& rsblk = lds_.rsblk;
  auto& ubuf = lds_.ubuf;
  auto& pbuf = lds_.pbuf;
  auto& wbuf = lds_.wbuf;
  auto& rbuf = lds_.rbuf;
  auto& psibuf = lds_.psibuf;
  auto& xbuf = lds_.xbuf;
  const int tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;  // lane: position in the wave (tiles, DPP, MFMA layout)
  auto xsync = [&]() { if (NW == 2) lds_barrier(); };             // (LDS-only wait: the tile prefetch stays in flight)
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int row = tid / LPR, seg = tid % LPR;
  const int Wc = JR + 2 * JC;                       // celerite rows
  const int W = Wc + (GEN ? P.J_general : 0);       // + general rows
  const bool writer = seg == 0;

  RowCoeffs rc{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (row < JR) {
    rc.u0 = P.a_real[(long)b * JR + row];
    rc.v0 = 1.0;
    rc.c = P.c_real[(long)b * JR + row];
  } else if (row < Wc) {
    const int j = (row - JR) >> 1;
    const double a = P.a_comp[(long)b * JC + j], bb = P.b_comp[(long)b * JC + j];
    if (((row - JR) & 1) == 0) { rc.uc = a; rc.us = bb; rc.vc = 1.0; }   // cholesky.h:143,145
    else                       { rc.uc = -bb; rc.us = a; rc.vs = 1.0; }  // cholesky.h:144,146
    rc.c = P.c_comp[(long)b * JC + j];
    rc.d = P.d_comp[(long)b * JC + j];
  }
  // K(0) = sum a_real + sum a_comp (+ jitter): the same summation order as the scan
  // kernels' Problem::diagonal (cholesky.h:98-100,120)
  double sum_ar = 0.0, sum_ac = 0.0;
  for (int j = 0; j < JR; ++j) sum_ar += P.a_real[(long)b * JR + j];
  for (int j = 0; j < JC; ++j) sum_ac += P.a_comp[(long)b * JC + j];
  const double jitter = P.jitter[b];

  const double* tp = P.t + b * P.t_stride;
  const double* dp = P.diag + b * P.diag_stride;
  const double* yp = P.y + b * P.y_stride;
  const int N = P.N;
  const bool gen = GEN && row >= Wc && row < W;
  const double* Ug = gen ? P.gen_U + b * P.gen_U_stride + (long)(row - Wc) * N : nullptr;
  const double* Vg = gen ? P.gen_V + b * P.gen_V_stride + (long)(row - Wc) * N : nullptr;
  const double* Ap = GEN ? P.gen_A + b * P.gen_A_stride : nullptr;
  // chunked replay: forced-exact runs, or the problems the conditioning record sent here (level 1)
  if (MODE == 0 && P.nchunk > 1 && !P.force_exact && (P.need_exact[b] != 1 || (P.defer_level1 && !P.seq_only))) return;
  if (MODE == 0 && P.seq_only && P.need_exact[b] < 2) return;  // sequential pass: level >= 2 only
  // (materialising runs, BatchParams::ends / fixup_steps: the fix-up pass recomputes the heads of the chunks c >= 1 from
  //  the state the previous chunk's replay reached at the boundary -- clr_batch_kernels.h, replay_kernel has the story)
  const bool fixup = MODE == 0 && P.fixup_steps > 0;
  // (the output check -- BatchParams::head_check -- is the same pass for the problems at level 3 only; the plain fix-up
  //  leaves level >= 2 alone: the sequential pass writes that problem's factor)
  const bool hcheck = fixup && P.head_check != 0;
  if (hcheck && P.need_exact[b] != 3) return;
  if (hcheck && chunk == 0) {  // (chunk 0 starts from the exact zero state: its replay stands, its end state moves on)
    for (int i = tid; i < START; i += 64 * NW) P.ends[(long)b * P.nchunk * START + i] = P.ends_in[(long)b * P.nchunk * START + i];
    return;
  }
  if (fixup && !hcheck && (chunk == 0 || P.need_exact[b] >= 2)) return;
  float hc_dd = 0.f, hc_dw = 0.f, hc_wm = 0.f;  // (output check) largest |D / D' - 1|, |w - w'|, |w'| against the entries in place
  const int n_lo = __builtin_amdgcn_readfirstlane(wide_chunk_begin(P, chunk));  // (wave-uniform by construction: keep the loop counters scalar)
  const int n_end = __builtin_amdgcn_readfirstlane(wide_chunk_begin(P, chunk + 1));
  const int n_hi = fixup ? (n_lo + P.fixup_steps < n_end ? n_lo + P.fixup_steps : n_end) : n_end;
  const long slot = (long)b * P.nchunk + chunk;

  double S[COLS];
  double f = 0.0, quad = 0.0;
#pragma unroll
  for (int c = 0; c < COLS; ++c) S[c] = 0.0;
  if (MODE == 0 && chunk > 0) {  // start state of this chunk (packed upper triangle | f), from the prefix phase
    const double* st = fixup ? (hcheck ? P.ends_in : P.ends) + (slot - 1) * START : P.starts + slot * START;
#pragma unroll
    for (int c = 0; c < COLS; ++c) S[c] = st[sym(row, seg * COLS + c)];
    f = st[SZ + row];
  }
  double AT[RID ? COLS : 1], Jm[(RID && !JMM) ? COLS : 1], eta = 0.0;
  mfma_acc_t Jacc[JMM ? NTW : 1];  // (JMM) the tiles (ti, tj), ti <= tj, of Jm's upper triangle, row by row (NW = 2: tile q of wave q % 2)
#pragma unroll
  for (int q_ = 0; q_ < (JMM ? NTW : 1); ++q_) Jacc[q_] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
  double dprod = 1.0;  // (LPWIN) product of the current block's pivots
  if (RID) {
#pragma unroll
    for (int c = 0; c < COLS; ++c) AT[c] = (seg * COLS + c == row) ? 1.0 : 0.0;
    if (!JMM) {
#pragma unroll
      for (int c = 0; c < COLS; ++c) Jm[c] = 0.0;
    }
  }
  LogProduct lp;
  lp.init();
  int flag = 0;
  double gam = 0.0;  // summarize: max a_n / D_n of the zero-start pivots (conditioning record)

  // 64-sample register tiles of the series (one coalesced 512-B load per array),
  // handed out with v_readlane; t needs two samples of look-ahead
  double tv, dv, yv, tv2;
  {
    const int m = n_lo + lane;
    tv = m < N ? tp[m] : 0.0;
    dv = m < N ? dp[m] : 0.0;
    dv = ((dv + sum_ar) + sum_ac) + jitter;  // K(0) of the tile's samples: the reference's summation order, once per tile
    if (GEN && m < N) dv += Ap[m];  // (the reference adds A last, cholesky.h:99)
    yv = m < N ? yp[m] : 0.0;
    tv2 = m + 64 < N ? tp[m + 64] : 0.0;
  }
  auto t_at = [&](int k) { return k < 64 ? lane_value(tv, k) : lane_value(tv2, k - 64); };  // k < 66
  // (LAZY) the steps of the tile, prepared once per 64 samples: lane k holds t(n0 + k + 2) - t(n0 + k + 1), the one
  // scalar a step still has to fetch (the rotation's step is the previous step's decay step; t itself is only needed
  // at the anchors) -- instead of two wave-uniform reads of t with their tile selects per step
  auto tile_steps = [&](int n0) {
    const int i1 = (lane + 1) & 63, i2 = (lane + 2) & 63;
    const double a1 = __shfl(tv, i1, 64), b1 = __shfl(tv2, i1, 64), a2 = __shfl(tv, i2, 64), b2 = __shfl(tv2, i2, 64);
    const double t1v = lane + 1 < 64 ? a1 : b1, t2v = lane + 2 < 64 ? a2 : b2;
    return (n0 + lane + 2 < N) ? t2v - t1v : 0.0;
  };
  double dxt = (LAZY && !FB) ? tile_steps(n_lo) : 0.0, dxprev = 0.0;
  // (GEN) the general rows' features of samples base .. base + GEN_PF - 1; gen_next() hands out the front one as the
  // row's constants and fetches the sample GEN_PF further on
  double gu[GEN ? GEN_PF : 1], gv[GEN ? GEN_PF : 1];
  int gbase = n_lo;
  if (GEN) {
#pragma unroll
    for (int k = 0; k < GEN_PF; ++k) {
      gu[k] = (gen && n_lo + k < N) ? Ug[n_lo + k] : 0.0;
      gv[k] = (gen && n_lo + k < N) ? Vg[n_lo + k] : 0.0;
    }
  }
  constexpr int GQ = GEN ? GEN_PF : 1;
  auto gen_next = [&]() {
    if (gen) { rc.u0 = gu[0]; rc.v0 = gv[0]; }
#pragma unroll
    for (int k = 0; k + 1 < GQ; ++k) { gu[k] = gu[k + 1]; gv[k] = gv[k + 1]; }
    const int m = gbase + GQ;
    gu[GQ - 1] = (gen && m < N) ? Ug[m] : 0.0;
    gv[GQ - 1] = (gen && m < N) ? Vg[m] : 0.0;
    ++gbase;
  };
  if (GEN && !FBA) gen_next();  // the chunk's first sample

  // features of the chunk's first sample
  double u, v, phi;
  double psi = 1.0, psinv = 1.0, phinv = 1.0, csr = 1.0, sdr = 0.0, tcur = t_at(0);  // (LAZY)
  // (FB) per lane: tl / tr = the time its Psi / its (cos, sin) pair stand at; tpre = the ring's next 64 times in flight
  double tl = 0.0, tr = 0.0, tpre = 0.0;
  FbConsts KF;
  if (FB) KF.init(NB > 2);
  auto t_clamped = [&](int m) { return tp[m < N ? m : N - 1]; };  // (past the end: the last time, i.e. steps of 0)
  const int bq = PR ? (row & 1) * LPR + seg : seg;  // this lane's sample within a batch
  // (GEN with the feature batch) this general row's U, V of the lane's sample in the NEXT batch
  double gq_u = (GEN && FBA && gen && n_lo + bq < N) ? Ug[n_lo + bq] : 0.0;
  double gq_v = (GEN && FBA && gen && n_lo + bq < N) ? Vg[n_lo + bq] : 0.0;
  // (PR) the term's own a, b (the row holds (a, b) as (uc, us) or (us, -uc): cholesky.h:143-146)
  const double ta = (row & 1) ? rc.us : rc.uc, tb = (row & 1) ? -rc.uc : rc.us;
  // (PR) 1 on the term's rows, 0 on padding rows: their v must stay 0 like their u (the pair form writes cos | sin for
  // every row pair; nonzero v on padding rows is contained in the padding entries of the element, but keep those clean)
  const double vone = (PR && row >= W) ? 0.0 : 1.0;
  auto feature_batch = [&](int m, bool anchor) {  // samples m .. m + NB - 1, one per lane of a row (PR: of a term)
    const int ms = m + bq;
    const double tm = tring[ms & 127];
    if constexpr (GEN) {  // a general row's constants of this sample (loaded a batch ago), the next batch's on their way
      if (gen) { rc.u0 = gq_u; rc.v0 = gq_v; }
      const int mn = ms + NB;
      gq_u = (gen && mn < N) ? Ug[mn] : 0.0;
      gq_v = (gen && mn < N) ? Vg[mn] : 0.0;
    }
    if constexpr (!LAZY) {  // (FBN) u~, v~ at t_ms and the decay to t_(ms + 1) (past the end the ring repeats the last time: phi = 1)
      if constexpr (PR) {  // one sincos and one exp for the term's two rows (row_features spelled out for the pair)
        double sd, cs;
        sincos_phase<FAST>(rc.d * tm, &sd, &cs);
        const double x = -rc.c * (tring[(ms + 1) & 127] - tm);
        const double ph = CLR_WAVE_ALL(fabs(x) < 0.0078125) ? exp_small(x) : exp(x);
        const int r0 = row & ~1, sl = ms & (NSLOT - 1);
        fub[sl][r0] = fma(ta, cs, tb * sd);
        fub[sl][r0 + 1] = fma(ta, sd, -(tb * cs));
        fvb[sl][r0] = vone * cs;
        fvb[sl][r0 + 1] = vone * sd;
        fpb[sl][r0] = ph;
        fpb[sl][r0 + 1] = ph;
      } else {
        double uu, vv, ph;
        row_features<FAST>(rc, tm, tring[(ms + 1) & 127] - tm, &uu, &vv, &ph);
        fub[ms & (NSLOT - 1)][row] = uu;
        fvb[ms & (NSLOT - 1)][row] = vv;
        fpb[ms & (NSLOT - 1)][row] = ph;
      }
      return;
    }
    // The series and Taylor steps above hold for |d dt| < NB 2^-5, |c dt| < NB 2^-7 -- every step of a densely sampled
    // series (the host's strict rule: GAPS == false, no test here).  GAPS: a lane whose interval is larger (an observing
    // gap) sends the WAVE through the full sincos / exp for this batch: the lazy flavour is then exact for any series
    // whose accumulated decay between two renormalisations stays representable (the host admits max c x max dx < 2:
    // Psi^-2 < e^256), and a gap costs one slow batch.  A flavour of its own: the slow path's constants and the two
    // compares cost the dense flavour 5 % when they share a kernel (profiles/r05m_wide_lazy_gaps.txt).
    const double ang = rc.d * (tm - tr), xdec = -rc.c * (tm - tl);
    const bool small = !GAPS || CLR_WAVE_ALL(fabs(ang) < NB * 0.03125 && fabs(xdec) < NB * 0.0078125);
    if (anchor || !small) {
      sincos_phase<FAST>(rc.d * tm, &sdr, &csr);
    } else {
      double sn, cn;
      small_sincos_nb<NB>(KF, ang, &sn, &cn);
      const double c0 = csr, s0 = sdr;
      csr = fma(c0, cn, -s0 * sn);
      sdr = fma(s0, cn, c0 * sn);
    }
    tr = tm;
    double e, einv;
    if (small) decay_pair_nb<NB>(KF, xdec, &e, &einv);
    else { e = exp(xdec); einv = exp(-xdec); }
    psi *= e;
    psinv *= einv;
    tl = tm;
    if constexpr (PR) {
      const int r0 = row & ~1;
      const double uc_ = fma(ta, csr, tb * sdr), us_ = fma(ta, sdr, -(tb * csr));
      fub[ms & (NSLOT - 1)][r0] = psi * uc_;
      fub[ms & (NSLOT - 1)][r0 + 1] = psi * us_;
      fvb[ms & (NSLOT - 1)][r0] = (vone * psinv) * csr;
      fvb[ms & (NSLOT - 1)][r0 + 1] = (vone * psinv) * sdr;
    } else {
      const double uu = fma(rc.uc, csr, fma(rc.us, sdr, rc.u0));
      const double vv = fma(rc.vc, csr, fma(rc.vs, sdr, rc.v0));
      fub[ms & (NSLOT - 1)][row] = psi * uu;
      fvb[ms & (NSLOT - 1)][row] = psinv * vv;
    }
  };
  if (FBA) {
    const int m = n_lo + lane;
    tring[m & 127] = t_clamped(m);
    tring[(m + 64) & 127] = t_clamped(m + 64);
    tpre = t_clamped(m + 128);
    tl = tr = tring[n_lo & 127];
    feature_batch(n_lo, true);
    u = v = phi = 0.0;
    xsync();  // (NW = 2: the other wave's rows of the first samples)
  } else if (LAZY) {
    sincos_phase<FAST>(rc.d * tcur, &sdr, &csr);
    u = fma(rc.uc, csr, fma(rc.us, sdr, rc.u0));
    v = fma(rc.vc, csr, fma(rc.vs, sdr, rc.v0));
    dxprev = n_lo + 1 < N ? t_at(1) - tcur : 0.0;
    decay_pair<LAZY>(-rc.c * dxprev, &phi, &phinv);
    if (writer) ubuf[n_lo & 1][row] = u;  // ubar = psi u with psi = 1
  } else {
    row_features<FAST>(rc, t_at(0), n_lo + 1 < N ? t_at(1) - t_at(0) : 0.0, &u, &v, &phi);
    if (writer) { ubuf[n_lo & 1][row] = u; pbuf[n_lo & 1][row] = phi; }
  }

  const double rsel_a = seg == 0 ? 1.0 : 0.0, rsel_b = seg == 0 ? 0.0 : -1.0;
  double* const rdst = (seg == 0 ? rblk : rsblk) + (JMM ? row : 0);  // (JMM, two lanes per row) where this lane parks r (first lane) or -r / D (second)
  double psiR = 1.0;  // (FB) the decay accumulated since the last renormalisation, up to t_(n + 1): set and used on renormalising steps
  double dmin = INFINITY;  // (LPWIN) smallest zero-start pivot of the samples >= 1
  // (LAZY) multiply the accumulated decay out of Sbar, Abar, fbar -- on renormalising steps
  auto renormalise = [&]() {
    if (writer) psibuf[row] = FB ? psiR : psi;
    xsync();
    // FB: the row's lanes carried their own Psi, equal up to rounding -- all of them take the writer's.  The
    // wave barrier is for the COMPILER: without it the other lanes' read of psibuf[row] is folded into the
    // else-arm of the writer's branch and issued before the store (legal for unsynchronised threads; the
    // hardware executes a wave's LDS operations in order)
    if (FB) __builtin_amdgcn_wave_barrier();
    const double prow = FB ? psibuf[row] : psi;
    const double2* qv = reinterpret_cast<const double2*>(&psibuf[seg * COLS]);
#pragma unroll
    for (int c = 0; c < COLS / 2; ++c) {
      const double2 pc = qv[c];
      S[2 * c] *= prow * pc.x;
      S[2 * c + 1] *= prow * pc.y;
      if (RID) {
        AT[2 * c] *= pc.x;
        AT[2 * c + 1] *= pc.y;
      }
    }
    f *= prow;
    if (!FB) {
      psi = 1.0;
      psinv = 1.0;
    }
  };
  for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
    const int nend = FBA ? opaque_s((n_hi - n0 < 64) ? n_hi - n0 : 64) : ((n_hi - n0 < 64) ? n_hi - n0 : 64);
    if (FBA) { dtile[lane] = dv; ytile[lane] = yv; }
    for (int k = 0; k < nend; ++k) {
      const int n = n0 + k, cur = n & 1;
      const double diag_n = FBA ? dtile[k] : lane_value(dv, k);
      const double y_n = FBA ? ytile[k] : lane_value(yv, k);

      // next sample's features (independent of the state): computed and published now
      double u1 = 0.0, v1 = 0.0, phi1 = 1.0, phinv1 = 1.0;
      // (round 5: every RN = 64 steps -- Psi >= exp(-1/2) there; the 16-step blocks of the pivots' product and of Jm's
      //  rank-16 updates do not depend on the base: r = Abar^T ubar is the true A^T u whatever Psi is)
      const bool renorm = LAZY && ((((n - n_lo) & (RN - 1)) == RN - 1) || n + 1 == n_hi);  // wave-uniform
      if constexpr (FB) {
        if (((n + 1 - n_lo) & (NB - 1)) == 0 || renorm) {  // (wave-uniform; the other steps touch none of this)
          if (renorm) {  // carry this lane's Psi to t_(n + 1) (forwards or, at a chunk's ragged end, backwards), new base there
            const double tb = tring[(n + 1) & 127];
            double e, einv;
            const double xr = -rc.c * (tb - tl);
            if (!GAPS || CLR_WAVE_ALL(fabs(xr) < NB * 0.0078125)) decay_pair_nb<NB>(KF, xr, &e, &einv);
            else e = exp(xr);  // (a gap inside the lane's interval)
            psiR = psi * e;
            psi = 1.0;
            psinv = 1.0;
            tl = tb;
          }
          if (((n + 1 - n_lo) & (NB - 1)) == 0 && n + 1 < n_hi) feature_batch(n + 1, ((n + 1 - n_lo) & (16 * NB - 1)) == 0);
        }
      } else if constexpr (FBN) {
        if (((n + 1 - n_lo) & (NB - 1)) == 0 && n + 1 < n_hi) feature_batch(n + 1, false);
        const double* pu = &fub[n & (NSLOT - 1)][row];  // this sample's u, v, phi of the row
        u = pu[0];
        v = pu[NSLOT * WMAX];
        phi = pu[2 * NSLOT * WMAX];
      } else if (n + 1 < N) {
        const double t1 = LAZY ? 0.0 : t_at(k + 1);
        const double dx1 = LAZY ? lane_value(dxt, k) : ((n + 2 < N) ? t_at(k + 2) - t1 : 0.0);
        if (GEN && !FBA) gen_next();  // the general rows' u0, v0 of sample n + 1
        if (LAZY) {
          if (((n + 1 - n_lo) & 15) == 0) {
            sincos_phase<FAST>(rc.d * t_at(k + 1), &sdr, &csr);  // anchor
          } else {  // rotate the row's (cos, sin) pair through d (t1 - t)
            const double dl = rc.d * dxprev, d2 = dl * dl;
            const double sn = dl * fma(d2, fma(d2, fma(d2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
            const double cn = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
            const double c0 = csr, s0 = sdr;
            csr = fma(c0, cn, -s0 * sn);
            sdr = fma(s0, cn, c0 * sn);
          }
          dxprev = dx1;
          u1 = fma(rc.uc, csr, fma(rc.us, sdr, rc.u0));
          v1 = fma(rc.vc, csr, fma(rc.vs, sdr, rc.v0));
          decay_pair<LAZY>(-rc.c * dx1, &phi1, &phinv1);
          // ubar of the next sample: Psi then includes this step's decay -- or is 1 after a renormalisation
          if (writer) ubuf[cur ^ 1][row] = renorm ? u1 : (psi * phi) * u1;
        } else {
          row_features<FAST>(rc, t1, dx1, &u1, &v1, &phi1);
          if (writer) { ubuf[cur ^ 1][row] = u1; pbuf[cur ^ 1][row] = phi1; }
        }
      }
      const double* ucols = FBA ? &fub[n & (NSLOT - 1)][seg * COLS] : &ubuf[cur][FBA ? 0 : seg * COLS];
      double ueff, veff;  // this row's entries of ubar (what the row of u in LDS holds) and vbar
      if constexpr (FB) {  // (one address, the vbar entry at a constant distance)
        const double* pu = &fub[n & (NSLOT - 1)][row];
        ueff = pu[0];
        veff = pu[NSLOT * WMAX];
      }
      else { ueff = LAZY ? psi * u : u; veff = LAZY ? psinv * v : v; }

      // q = S u and (summarize) r = A^T u: own columns, then across the row's lanes
      double q = 0.0, r = 0.0;
      if constexpr (COLS >= 32) {
        // width 64: one lane owns a whole row -- a single accumulator would be a chain of 64 dependent FMAs per dot
        // product, and the wave is alone on its SIMD (nothing hides the latency): four partial sums each
        double qa[4] = {0.0, 0.0, 0.0, 0.0}, ra[4] = {0.0, 0.0, 0.0, 0.0};
        const double2* uv = reinterpret_cast<const double2*>(ucols);
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 uu = uv[c];
          qa[(2 * c) & 3] = fma(S[2 * c], uu.x, qa[(2 * c) & 3]);
          qa[(2 * c + 1) & 3] = fma(S[2 * c + 1], uu.y, qa[(2 * c + 1) & 3]);
          if (RID) {
            ra[(2 * c) & 3] = fma(AT[2 * c], uu.x, ra[(2 * c) & 3]);
            ra[(2 * c + 1) & 3] = fma(AT[2 * c + 1], uu.y, ra[(2 * c + 1) & 3]);
          }
        }
        q = (qa[0] + qa[1]) + (qa[2] + qa[3]);
        if (RID) r = (ra[0] + ra[1]) + (ra[2] + ra[3]);
      } else {
        const double2* uv = reinterpret_cast<const double2*>(ucols);
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 uu = uv[c];
          q = fma(S[2 * c], uu.x, q);
          q = fma(S[2 * c + 1], uu.y, q);
          if (RID) {
            r = fma(AT[2 * c], uu.x, r);
            r = fma(AT[2 * c + 1], uu.y, r);
          }
        }
      }
      if (LPR >= 2) { q = dpp_add<DPP_QUAD_XOR1>(q); if (RID) r = dpp_add<DPP_QUAD_XOR1>(r); }
      if (LPR >= 4) { q = dpp_add<DPP_QUAD_XOR2>(q); if (RID) r = dpp_add<DPP_QUAD_XOR2>(r); }
      double s, ub;
#ifndef CLR_WIDE_PERMLANE_SUMS
#define CLR_WIDE_PERMLANE_SUMS 1  // the sums completed with v_permlane16/32_swap and kept in vector registers
#endif
      if constexpr (PACKED && CLR_WIDE_PERMLANE_SUMS) row_sum2_all<LPR>(ueff * (seg == 0 ? q : f), seg, &s, &ub);
      else if constexpr (PACKED) row_sum2<LPR>(ueff * (seg == 0 ? q : f), seg, &s, &ub);
      else { s = row_sum_all<LPR>(ueff * q); ub = row_sum_all<LPR>(ueff * f); }  // (one lane per row: width 64)
      if (NW == 2) {  // the other wave's rows: partial sums through LDS, added in the same order by both waves
        if (lane == 0) { xbuf[2 * wvi] = s; xbuf[2 * wvi + 1] = ub; }
        xsync();
        s = xbuf[0] + xbuf[2];
        ub = xbuf[1] + xbuf[3];
      }
      const double D = diag_n - s;  // (diag_n: the tile already holds K(0) = ((diag + sum a_real) + sum a_comp) + jitter)
      const double invD = (MODE == 1) ? recip_fast(D) : 1.0 / D;  // (the replay writes W = z / D into the factor: IEEE)
      const double x = y_n - ub;
      // replay: the reference's test (cholesky.h:176; sample 0 is never checked); summarize: a
      // zero-start pivot <= 0 sends the problem to the replay (as in summarize_chunk)
      if (LPWIN) {  // the smallest pivot, tested once per block (a NaN pivot poisons the block's product, tested there too)
        // (sample 0 is never checked -- cholesky.h:176 -- and only the first chunk, the body without riders, holds it)
        if (RIDERS) dmin = min3(dmin, D);
        else if (n >= 1) dmin = fmin(dmin, D);
      } else if (n >= 1 && (MODE == 1 ? !(D > 0.0) : D < 0.0)) flag = 1;
      if (LPWIN) dprod *= D; else lp.mul(D);
      const double xs = x * invD;
      quad = fma(x, xs, quad);
      if (MODE == 1) gam = max_abs(gam, diag_n * invD);

      const double z = veff - q;
      const double w = z * invD;
      if (writer) {
        wbuf[row] = LAZY ? w : phi * w;
        if (RID && !JMM) rbuf[row] = r;
      }
      if (JMM) {  // this step's r (first lane of the row) and -r / D (second lane), for the block's rank-16 update
        if (LPR >= 2) {  // r (first lane) | -r / D (second lane) = r * (ra + rb / D) with per-lane constants: no selects
          rdst[((n - n_lo) & 15) * WMAX] = r * fma(invD, rsel_b, rsel_a);
        } else {  // (width 64: one lane per row writes both)
          rblk[((n - n_lo) & 15) * WMAX + row] = r;
          rsblk[((n - n_lo) & 15) * WMAX + row] = -(r * invD);
        }
      }
      xsync();  // (NW = 2: the step's w -- and r -- of BOTH waves' rows, before the rank-1 updates read them)
      // (one wave: program order is enough for the hardware; this pins it for the compiler too -- the rank-1 updates
      //  below read wbuf / rbuf entries other lanes wrote inside the `writer` arm above)
      if (NW == 1) __builtin_amdgcn_wave_barrier();
      if (MODE == 0 && P.wide_materialize && writer && row < W) {
        // the factor in the reference's storage, element (j, n) at [j + W n]: W[:, n], D[n],
        // u[:, n - 1] = U~(t_n), phi[:, n] = decay n -> n + 1 (cholesky.h:131-151, :177-178)
        const long Wl = W;
        if (hcheck) {  // what the first replay wrote (a NaN difference counts as infinite)
          const double wo = P.W[(long)b * Wl * N + Wl * n + row];
          const float dw = (float)fabs(w - wo), wm = (float)fabs(wo);
          if (!(dw <= hc_dw)) hc_dw = (dw != dw) ? INFINITY : dw;
          if (!(wm <= hc_wm)) hc_wm = (wm != wm) ? INFINITY : wm;
          if (row == 0) {
            const double Do = P.D[(long)b * N + n];
            const float dd = (float)fabs((D - Do) / Do);
            if (!(dd <= hc_dd)) hc_dd = (dd != dd) ? INFINITY : dd;
          }
        }
        P.W[(long)b * Wl * N + Wl * n + row] = w;
        if (row == 0) P.D[(long)b * N + n] = D;
        if (n >= 1) P.u[(long)b * Wl * (N - 1) + Wl * (n - 1) + row] = u;
        if (n + 1 < N) P.phi[(long)b * Wl * (N - 1) + Wl * n + row] = phi;
      }
      if (LAZY) {
        const double2* wv = reinterpret_cast<const double2*>(&wbuf[seg * COLS]);
        const double2* rv = reinterpret_cast<const double2*>(&rbuf[(RID && !JMM) ? seg * COLS : 0]);
        const double rs = r * invD;
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 pw = wv[c];
          S[2 * c] = fma(z, pw.x, S[2 * c]);
          S[2 * c + 1] = fma(z, pw.y, S[2 * c + 1]);
          if (RID) {
            AT[2 * c] = fma(-pw.x, r, AT[2 * c]);
            AT[2 * c + 1] = fma(-pw.y, r, AT[2 * c + 1]);
          }
          if (RID && !JMM) {
            const double2 rr = rv[c];
            Jm[2 * c] = fma(-rs, rr.x, Jm[2 * c]);
            Jm[2 * c + 1] = fma(-rs, rr.y, Jm[2 * c + 1]);
          }
        }
      } else {
        const double2* pv = reinterpret_cast<const double2*>(FBA ? &fpb[n & (NSLOT - 1)][seg * COLS] : &pbuf[cur][FBA ? 0 : seg * COLS]);
        const double2* wv = reinterpret_cast<const double2*>(&wbuf[seg * COLS]);
        const double2* rv = reinterpret_cast<const double2*>(&rbuf[(RID && !JMM) ? seg * COLS : 0]);
        const double zr = phi * z, rs = r * invD;
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
          const double2 pk = pv[c], pw = wv[c];
          S[2 * c] = fma(zr, pw.x, (phi * pk.x) * S[2 * c]);
          S[2 * c + 1] = fma(zr, pw.y, (phi * pk.y) * S[2 * c + 1]);
          if (RID) {
            const double2 rr = rv[c];
            AT[2 * c] = fma(-pw.x, r, pk.x * AT[2 * c]);
            AT[2 * c + 1] = fma(-pw.y, r, pk.y * AT[2 * c + 1]);
            Jm[2 * c] = fma(-rs, rr.x, Jm[2 * c]);
            Jm[2 * c + 1] = fma(-rs, rr.y, Jm[2 * c + 1]);
          }
        }
      }
      if (RID) eta = fma(-r, xs, eta);
      if (LAZY) {
        f = fma(w, x, f);  // fbar
        if (!FB) {
          psi *= phi;      // Psi now includes this step's decay
          psinv *= phinv;
        }
        // (FB: the test again from an opaque copy of n -- carried across the step the flag costs a v_cndmask and a v_cmp)
        const int n2 = FB ? opaque_s(n) : n;
        const bool renorm_now = FB ? ((((n2 - n_lo) & (RN - 1)) == RN - 1) || n2 + 1 == n_hi) : renorm;
        const bool block_end = RN == 16 ? renorm_now : ((((n2 - n_lo) & 15) == 15) || n2 + 1 == n_hi);
        if (block_end) {
          if (LPWIN) {     // the block's pivots: one frexp for (at most) 16 of them
            if (!(dprod > 1e-250 && dprod < 1e250) || !(dmin > 0.0)) flag = 1;  // (over/underflow or NaN; a pivot <= 0)
            lp.mul_window(dprod);
            lp.renorm();
            dprod = 1.0;
          }
          if (JMM) {       // Jm += R (-D^-1 R)^T over the block's steps
            const int cnt = ((n - n_lo) & 15) + 1;
            if (cnt < 16) {  // the chunk's last, shorter block: the unused steps contribute nothing
              for (int idx = cnt * WMAX + tid; idx < 16 * WMAX; idx += 64 * NW) { rblk[idx] = 0.0; rsblk[idx] = 0.0; }
              xsync();
            }
            const int lm = lane & 15, lk = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              double av[NTL], bv[NTL];
#pragma unroll
              for (int ti = 0; ti < NTL; ++ti) {
                av[ti] = rblk[(4 * ks + lk) * WMAX + 16 * ti + lm];
                bv[ti] = rsblk[(4 * ks + lk) * WMAX + 16 * ti + lm];
              }
              int q_ = 0;
#pragma unroll
              for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
                for (int tj = ti; tj < NTL; ++tj, ++q_) {
                  if (NW == 1) Jacc[q_] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bv[tj], Jacc[q_], 0, 0, 0);
                  else if ((q_ & 1) == wvi) Jacc[q_ >> 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bv[tj], Jacc[q_ >> 1], 0, 0, 0);
                }
              }
            }
          }
          if (RN == 16) renormalise();
        }
        if (RN != 16 && renorm_now) renormalise();
        phinv = phinv1;
      } else {
        f = phi * (f + w * x);
        // NW = 2 without the feature slots (general terms): the rank-1 update above read pbuf[cur], the buffer the NEXT
        // step's features go to -- a wave that is one step ahead must not write it yet (found on 64 x 1e5 x width 64:
        // 1e-7 deviations of the two-wave kernel's plain flavour; the slots of FBN are written a full ring ahead)
        if (NW == 2 && !FBA) xsync();
      }
      if (!FBN) { u = u1; v = v1; phi = phi1; }
    }
    // next tile of the series
    if (FBA) {  // the ring's next 64 times: loaded a tile ago (before this tile's loads are issued: no wait on them),
               // they replace the samples just processed
      tring[(n0 + 128 + lane) & 127] = tpre;
      tpre = t_clamped(n0 + 192 + lane);
    }
    const int m = n0 + 64 + lane;
    tv = tv2;
    dv = m < N ? dp[m] : 0.0;
    dv = ((dv + sum_ar) + sum_ac) + jitter;
    if (GEN && m < N) dv += Ap[m];
    yv = m < N ? yp[m] : 0.0;
    tv2 = m + 64 < N ? tp[m + 64] : 0.0;
    if (LAZY && !FB) dxt = tile_steps(n0 + 64);
  }

  if (MODE == 1) {  // the element, in the narrow kernels' layout at width J = WMAX
    double* e = P.elems + slot * ELEM;
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = seg * COLS + c;
      e[col * J + row] = RID ? AT[c] : 0.0;                      // A[col][row] = AT[row][col]
      if (row <= col) {
        e[J * J + J + tri(row, col)] = S[c];                     // C, packed upper triangle
        if (!JMM) e[J * J + J + SZ + J + tri(row, col)] = RID ? Jm[c] : 0.0;  // Jm
      }
    }
    if (JMM) {  // accumulator layout of v_mfma_f64_16x16x4: register r of tile (ti, tj) is entry (16 ti + (lane >> 4) + 4 r, 16 tj + (lane & 15))
      const int lm = lane & 15, lk = lane >> 4;
      int q_ = 0;
#pragma unroll
      for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
        for (int tj = ti; tj < NTL; ++tj, ++q_) {
          if (NW == 2 && (q_ & 1) != wvi) continue;  // (the other wave's tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + lk + 4 * r, j = 16 * tj + lm;
            if (i <= j) e[J * J + J + SZ + J + tri(i, j)] = Jacc[NW == 2 ? (q_ >> 1) : q_][r];
          }
        }
      }
    }
    if (writer) {
      e[J * J + row] = f;                                        // b (the zero-start f)
      e[J * J + J + SZ + row] = eta;
    }
    if (tid == 0) {  // zero-start sums: correct_kernel<WMAX> turns them into the true contributions
      P.part[slot * 2 + 0] = lp.log_value();
      P.part[slot * 2 + 1] = quad;
      P.flags[slot] = flag;
      if (P.cond) { P.cond[slot * 3 + 0] = gam; P.cond[slot * 3 + 1] = 1.0; P.cond[slot * 3 + 2] = 0.0; }
    }
    return;
  }
  if (fixup) {  // (factor entries only: sums, flags and the record stand)
    if (hcheck && P.cond) {  // the output mismatch of this chunk, where the replay had left its end-state residual
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        hc_dd = fmaxf(hc_dd, __shfl_xor(hc_dd, m, 64));
        hc_dw = fmaxf(hc_dw, __shfl_xor(hc_dw, m, 64));
        hc_wm = fmaxf(hc_wm, __shfl_xor(hc_wm, m, 64));
      }
      if (NW == 2) {  // the other wave's rows
        xsync();
        if (lane == 0) { xbuf[4 * wvi] = hc_dd; xbuf[4 * wvi + 1] = hc_dw; xbuf[4 * wvi + 2] = hc_wm; }
        xsync();
        hc_dd = fmaxf((float)xbuf[0], (float)xbuf[4]); hc_dw = fmaxf((float)xbuf[1], (float)xbuf[5]); hc_wm = fmaxf((float)xbuf[2], (float)xbuf[6]);
      }
      if (tid == 0) {
        const double rw = (hc_wm > 0.f) ? (double)hc_dw / (double)hc_wm : (hc_dw == 0.f ? 0.0 : (double)INFINITY);
        P.cond[slot * 3 + 2] = fmax((double)hc_dd, rw);
      }
    }
    if (!hcheck) return;  // (the output check is a whole replay: its end state, sums and flags replace the first one's)
  }
  if (MODE == 0 && P.nchunk > 1 && P.ends) {  // the state at the chunk's end, for the fix-up pass
    double* e = P.ends + slot * START;
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = seg * COLS + c;
      if (row <= col && col < WMAX) e[sym(row, col)] = S[c];
    }
    if (writer) e[SZ + row] = f;
  }
  if (MODE == 0 && P.nchunk > 1 && P.cond && !hcheck) {
    // end state of this chunk against the scanned start state of the next one (as replay_kernel)
    double dp = 0.0, pm = 0.0, df = 0.0, fm = 0.0;
    if (chunk + 1 < P.nchunk) {
      const double* nx = P.starts + (slot + 1) * START;
#pragma unroll
      for (int c = 0; c < COLS; ++c) {
        const double ref = nx[sym(row, seg * COLS + c)];
        pm = fmax(pm, fabs(ref));
        dp = fmax(dp, fabs(ref - S[c]));
      }
      fm = fabs(nx[SZ + row]);
      df = fabs(nx[SZ + row] - f);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      dp = fmax(dp, __shfl_xor(dp, m, 64)); pm = fmax(pm, __shfl_xor(pm, m, 64));
      df = fmax(df, __shfl_xor(df, m, 64)); fm = fmax(fm, __shfl_xor(fm, m, 64));
    }
    if (NW == 2) {  // the other wave's rows
      xsync();
      if (lane == 0) { xbuf[4 * wvi] = dp; xbuf[4 * wvi + 1] = pm; xbuf[4 * wvi + 2] = df; xbuf[4 * wvi + 3] = fm; }
      xsync();
      dp = fmax(xbuf[0], xbuf[4]); pm = fmax(xbuf[1], xbuf[5]); df = fmax(xbuf[2], xbuf[6]); fm = fmax(xbuf[3], xbuf[7]);
    }
    if (tid == 0) {
      double res = (pm > 0.0) ? dp / pm : (dp == 0.0 ? 0.0 : INFINITY);
      if (fm > 0.0 && !P.logdet_only) res = fmax(res, df / fm);
      P.cond[slot * 3 + 2] = (chunk + 1 < P.nchunk) ? res : 0.0;
    }
  }
  if (tid == 0) {
    const double ld = lp.log_value();
    if (P.nchunk > 1) {  // partial sums of this chunk; finalize_kernel adds them up
      P.partx[slot * 2 + 0] = ld;
      P.partx[slot * 2 + 1] = quad;
      P.flagsx[slot] = flag;
    } else if (flag) {  // celerite::linalg_exception (cholesky.h:176); quiet => -inf (celerite.py:205-208)
      P.out_status[b] = CLR_NOT_POSITIVE_DEFINITE;
      P.out_ll[b] = -INFINITY;
      P.out_logdet[b] = NAN;
      P.out_quad[b] = NAN;
    } else {
      P.out_status[b] = CLR_OK;
      P.out_logdet[b] = ld;
      P.out_quad[b] = quad;
      P.out_ll[b] = combine_loglike(ld, quad, N);
    }
  }
}

template <int WMAX, bool FAST, int MODE, bool LAZY = false, bool GEN = false, bool PAIRED = false, bool GAPS = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) wide_scan_kernel(const BatchParams P, int JR, int JC) {
  if (MODE == 1 && blockIdx.x == 0) wide_scan_body<WMAX, FAST, MODE, LAZY, false, GEN, 1, PAIRED, GAPS>(P, JR, JC);
  else wide_scan_body<WMAX, FAST, MODE, LAZY, true, GEN, 1, PAIRED, GAPS>(P, JR, JC);
}
// Round 5: the summarize flavour at widths 33..64 (one lane per row, 64 columns each).  S and A^T alone are 2 x 64 doubles
// = 256 registers per lane, Jm sits in the matrix cores' accumulators (10 tiles x 4 doubles: the lazy flavour) or in 64
// more doubles: ONE wave per SIMD (512 registers), where the narrower kernels run two.
template <bool FAST, bool LAZY, bool GEN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) wide_summarize64_kernel(const BatchParams P, int JR, int JC) {
  if (blockIdx.x == 0) wide_scan_body<64, FAST, 1, LAZY, false, GEN>(P, JR, JC);
  else wide_scan_body<64, FAST, 1, LAZY, true, GEN>(P, JR, JC);
}
// ... and TWO waves per (problem, chunk): two lanes per row, 32 columns per lane (wide_scan_body, NW = 2) -- the state fits
// the architectural registers, two waves per SIMD, both the summarize (MODE 1) and the replay / sequential sweep (MODE 0).
template <bool FAST, int MODE, bool LAZY, bool GEN, bool PAIRED = false, bool GAPS = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) wide_scan64x2_kernel(const BatchParams P, int JR, int JC) {
  if (MODE == 1 && blockIdx.x == 0) wide_scan_body<64, FAST, MODE, LAZY, false, GEN, 2, PAIRED, GAPS>(P, JR, JC);
  else wide_scan_body<64, FAST, MODE, LAZY, true, GEN, 2, PAIRED, GAPS>(P, JR, JC);
}

#ifdef CLR_WIDE_SCAN32_ONLY
// ---------------------------------------------------------------------------
// wide_scan32.hip: this file compiled a SECOND time for the lazy summarize flavours at the padded width 32 alone
// (BASELINE configs[4]'s dominant kernel), with -mllvm -amdgpu-sched-strategy=max-ilp (Makefile).  Measured, A/B builds in
// one GPU call (profiles/r06zj_wide_sched_ab.txt, r06zl_wide_ilp_other_widths.txt): that strategy takes this kernel from
// 9.52-9.56 to 9.20-9.22 ms (same bits), but costs wide_correct_kernel 0.36 -> 0.48 ms, the 16-wide prefix 0.43 -> 0.54 ms
// and the width-64 summarize 41.7 -> 45.0 ms when the whole unit is built with it -- hence a unit of its own.
// ---------------------------------------------------------------------------
}  // namespace
void launch_wide_scan32_lazy(const BatchParams& P, int JR, int JC, bool paired, bool gaps, hipStream_t s) {
  const dim3 grid(P.nchunk, P.B);
  if (P.fast_trig) {
    if (paired && gaps) hipLaunchKernelGGL((wide_scan_kernel<32, true, 1, true, false, true, true>), grid, dim3(64), 0, s, P, JR, JC);
    else if (paired) hipLaunchKernelGGL((wide_scan_kernel<32, true, 1, true, false, true, false>), grid, dim3(64), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan_kernel<32, true, 1, true, false, false, true>), grid, dim3(64), 0, s, P, JR, JC);
  } else {
    if (paired && gaps) hipLaunchKernelGGL((wide_scan_kernel<32, false, 1, true, false, true, true>), grid, dim3(64), 0, s, P, JR, JC);
    else if (paired) hipLaunchKernelGGL((wide_scan_kernel<32, false, 1, true, false, true, false>), grid, dim3(64), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan_kernel<32, false, 1, true, false, false, true>), grid, dim3(64), 0, s, P, JR, JC);
  }
}
}  // namespace clr
#else
// ---------------------------------------------------------------------------
// correct at the padded widths 16 / 32: chunk_update + pd_certificate of clr_core.h (same
// formulas, same thresholds), one WAVE per (problem, chunk) with the matrices in LDS -- the
// single-lane form keeps 2080 doubles per lane in scratch at width 32 and was measured at
// 24 ms for 2048 chunks, slower than the replay pass it is meant to replace.
//   T = [ I + P Jm | P | f + P eta ]  --Gauss-Jordan, partial pivoting-->  [ . | G | g ], det
//   log det correction = log det ;  quad correction = 2 eta.f - f.Jm f + w.G w , w = Jm f - eta
//   certificate: F F^T = -Jm + delta I ; smallest Cholesky pivot of I - F^T P F > 1e-5
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int J>
// (round 4: 256 threads per (problem, chunk) -- the tableau / matrix sweeps are strided over four waves, the wave-uniform
//  scalars (pivot search, determinants, certificate pivots) are computed by every wave from the same LDS data, and the
//  per-row sections and the final reduction belong to the first wave: 0.83 -> 0.3 ms for BASELINE config 4)
__global__ void __launch_bounds__(256) wide_correct_kernel(const BatchParams P) {
  constexpr int SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  constexpr int NC = 2 * J + 1, LD = J + 1, LT = NC + 1;  // padded leading dimensions
  __shared__ double Pm[J * LD], Jmm[J * LD], Sm[J * LD], PF[J * LD], T[J * LT], fv[J], ev[J], wv[J];
  constexpr int NT = 256;
  const int tid = threadIdx.x, lane = tid & 63;  // lane: position in the wave (wave-uniform searches); tid < J: row owner
  const long slot = blockIdx.x;
  const int b = (int)(slot / P.nchunk), c = (int)(slot % P.nchunk);
  if (tid == 0 && P.flags[slot]) atomicOr(P.need_exact + b, 2);  // a zero-start pivot <= 0 (summarize)
  if (c == 0) {  // the first chunk starts from the zero state: nothing to correct
    if (tid == 0 && P.egerr) P.egerr[slot] = 0.0;
    return;
  }
  const double* st = P.starts + slot * START;
  const double* E = P.elems + slot * ELEM;
  const double* eta = E + J * J + J + SZ;
  const double* Jm = eta + J;
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    Pm[i * LD + j] = st[sym(i, j)];
    Jmm[i * LD + j] = Jm[sym(i, j)];
  }
  if (tid < J) { fv[tid] = st[SZ + lane]; ev[tid] = eta[tid]; }
  __syncthreads();

  // T = [ I + P Jm | P | f + P eta ]
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    double acc = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < J; ++k) acc += Pm[i * LD + k] * Jmm[k * LD + j];
    T[i * LT + j] = acc;
    T[i * LT + J + j] = Pm[i * LD + j];
  }
  if (tid < J) {
    double h = fv[tid];
    for (int j = 0; j < J; ++j) h += Pm[tid * LD + j] * ev[j];
    T[tid * LT + 2 * J] = h;
  }
  __syncthreads();

  double det = 1.0;
  for (int col = 0; col < J; ++col) {
    // pivot row: the largest |T[i][col]|, i >= col (first one on ties, as the single-lane scan)
    double best = (lane < J && lane >= col) ? fabs(T[lane * LT + col]) : -1.0;
    int piv = lane;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const double ob = __shfl_xor(best, m, 64);
      const int op = __shfl_xor(piv, m, 64);
      const bool take = ob > best || (ob == best && op < piv);
      best = take ? ob : best;
      piv = take ? op : piv;
    }
    __syncthreads();  // (every wave has finished its search of column `col`)
    if (piv != col) {
      for (int cc = tid; cc < NC; cc += NT) {
        const double a = T[col * LT + cc], bb = T[piv * LT + cc];
        T[col * LT + cc] = bb;
        T[piv * LT + cc] = a;
      }
    }
    __syncthreads();
    const double p = T[col * LT + col];
    det *= (piv != col) ? -p : p;
    const double inv = 1.0 / p;
    __syncthreads();
    for (int cc = tid; cc < NC; cc += NT)
      if (cc > col) T[col * LT + cc] *= inv;
    __syncthreads();
    for (int idx = tid; idx < J * NC; idx += NT) {
      const int i = idx / NC, cc = idx % NC;
      if (i != col && cc > col) T[i * LT + cc] -= T[i * LT + col] * T[col * LT + cc];
    }
    __syncthreads();
  }
  // T[i][J + j] = G[i][j] (symmetrised below), T[i][2J] = g[i]

  // measured accuracy of G (chunk_update's eg_out at this width, same two probe vectors): lane i < J owns row i;
  // the vectors cross the wave through wv / ev-sized LDS scratch (fv, ev, wv are rewritten below only after use)
  double eg = 0.0;
  if (P.egerr) {
    __shared__ double pa[J], pb[J];
    for (int probe = 0; probe < 2; ++probe) {
      double gz = 0.0, r = 0.0;
      if (tid < J) {
        for (int k = 0; k < J; ++k) gz += (probe && (k & 1)) ? -T[tid * LT + J + k] : T[tid * LT + J + k];
        pa[tid] = gz;
      }
      __syncthreads();
      if (tid < J) {  // t1 = Jm (G z)
        double acc = 0.0;
        for (int k = 0; k < J; ++k) acc += Jmm[tid * LD + k] * pa[k];
        pb[tid] = acc;
      }
      __syncthreads();
      if (tid < J) {  // r = P (z - t1) - G z
        double acc = -gz;
        for (int k = 0; k < J; ++k) acc += Pm[tid * LD + k] * (((probe && (k & 1)) ? -1.0 : 1.0) - pb[k]);
        r = acc;
        pa[tid] = r;
      }
      __syncthreads();
      if (tid < J) {  // t1 = Jm r
        double acc = 0.0;
        for (int k = 0; k < J; ++k) acc += Jmm[tid * LD + k] * pa[k];
        pb[tid] = acc;
      }
      __syncthreads();
      double dg = 0.0;
      if (tid < J) {  // dg = r - G t1
        double acc = r;
        for (int k = 0; k < J; ++k) acc -= T[tid * LT + J + k] * pb[k];
        dg = (acc != acc) ? INFINITY : fabs(acc);
      }
      double gmax = (tid < J) ? fabs(gz) : 0.0;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        dg = fmax(dg, __shfl_xor(dg, m, 64));
        gmax = fmax(gmax, __shfl_xor(gmax, m, 64));
      }
      const double e = (gmax > 0.0) ? dg / gmax : (dg == 0.0 ? 0.0 : INFINITY);
      eg = (e > eg || e != e) ? e : eg;
      __syncthreads();
    }
  }

  double ef = 0.0, fJf = 0.0, wGw = 0.0;
  if (tid < J) {
    double acc = 0.0;
    for (int k = 0; k < J; ++k) acc += Jmm[tid * LD + k] * fv[k];
    wv[tid] = acc - ev[tid];
    ef = ev[tid] * fv[tid];
    fJf = fv[tid] * acc;
  }
  __syncthreads();
  if (tid < J) {
    double acc = 0.0;
    for (int k = 0; k < J; ++k) acc += 0.5 * (T[tid * LT + J + k] + T[k * LT + J + lane]) * wv[k];
    wGw = wv[tid] * acc;
  }
  ef = wave_sum64(ef);
  fJf = wave_sum64(fJf);
  wGw = wave_sum64(wGw);

  // certificate: F F^T = N + delta I (N = -Jm), then the Cholesky pivots of I - F^T P F
  double nmax = (lane < J) ? -Jmm[lane * LD + lane] : 0.0;  // (every wave: delta is needed by all threads)
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) nmax = fmax(nmax, __shfl_xor(nmax, m, 64));
  const double delta = 4e-13 * nmax;
  for (int idx = tid; idx < J * J; idx += NT) {
    const int i = idx / J, j = idx % J;
    Sm[i * LD + j] = -Jmm[i * LD + j] + ((i == j) ? delta : 0.0);
  }
  __syncthreads();
  for (int k = 0; k < J; ++k) {  // in place: column k of F replaces column k of S (rows >= k)
    const double rs = 1.0 / sqrt(Sm[k * LD + k]);
    __syncthreads();
    if (tid < J) Sm[tid * LD + k] = (tid >= k) ? Sm[tid * LD + k] * rs : 0.0;
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (j > k && i >= j) {
        Sm[i * LD + j] -= Sm[i * LD + k] * Sm[j * LD + k];
      }
    }
    __syncthreads();
  }
  // (F is the lower triangle of Sm; the strict upper triangle still holds N)
  for (int idx = tid; idx < J * J; idx += NT) {  // PF = P F
    const int i = idx / J, k = idx % J;
    double acc = 0.0;
    for (int m = k; m < J; ++m) acc += Pm[i * LD + m] * Sm[m * LD + k];
    PF[i * LD + k] = acc;
  }
  __syncthreads();
  double* Em = T;  // T is no longer needed: E = I - F^T (P F), lower triangle, leading dimension LT
  for (int idx = tid; idx < J * J; idx += NT) {
    const int j = idx / J, k = idx % J;
    if (k <= j) {
      double acc = (j == k) ? 1.0 : 0.0;
      for (int i = k; i < J; ++i) acc -= Sm[i * LD + k] * PF[i * LD + j];
      Em[j * LT + k] = acc;
    }
  }
  __syncthreads();
  double mu = 1.0;
  bool broke = false;
  for (int k = 0; k < J; ++k) {
    const double d = Em[k * LT + k];
    if (!(d > 0.0)) broke = true;
    mu = (d < mu) ? d : mu;
    const double rs = 1.0 / sqrt(d);
    __syncthreads();
    if (tid < J && tid >= k) Em[tid * LT + k] *= rs;
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (j > k && i >= j) Em[i * LT + j] -= Em[i * LT + k] * Em[j * LT + k];
    }
    __syncthreads();
  }
  if (broke) mu = -1.0;

  if (tid == 0) {
    int bad = 0;
    if (!(mu > 1e-5)) bad = 1;
    if (!(det > 0.0)) bad = 1;
    const double ld0 = P.part[slot * 2 + 0], q0 = P.part[slot * 2 + 1];
    const double q = 2.0 * ef - fJf + wGw;
    const double ld = log(det);
    const double err = J * 2.2e-16 / mu;  // rounding-error estimate of the corrections
    // (err is summed over the problem's chunks and held against the problem's log det / quadratic form by decide_kernel:
    //  chunk_update's err_out, clr_core.h)
    if ((!P.logdet_only && !isfinite(q)) || !isfinite(ld)) bad = 1;
    P.part[slot * 2 + 0] = ld0 + ld;
    P.part[slot * 2 + 1] = q0 + q;
    if (P.cond) { P.cond[slot * 3 + 1] = mu; P.cond[slot * 3 + 2] = P.logdet_only ? 0.0 : err * fabs(wGw); }
    if (P.egerr) P.egerr[slot] = eg;
    if (bad) {
      P.flags[slot] |= 2;
      atomicOr(P.need_exact + b, 2);
    }
  }
}

// ---------------------------------------------------------------------------
// fp32 probe (BASELINE config 5: "fp32 vs fp64 tolerance"): the sequential sweep of wide_scan_kernel
// MODE 0 with the STATE and every per-step operation in float -- S, f, q, D, z, w, x, the LDS
// exchanges and the DPP reductions -- while the features (absolute phases d t, decays) are evaluated
// in fp64 and rounded, and log det / the quadratic form are accumulated in fp64 from the float pivots
// (the arrangement tools/fp32_tolerance.py emulates in NumPy).  A measurement, not a product path:
// it answers on the device what a float state costs in accuracy and buys in time at width <= 32.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
  return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value_f32(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
template <int LPR>
__device__ __forceinline__ float row_sum_f32(float v) {
  if (LPR < 2) v = dpp_add_f32<DPP_QUAD_XOR1>(v);
  if (LPR < 4) v = dpp_add_f32<DPP_QUAD_XOR2>(v);
  v = dpp_add_f32<DPP_HALF_MIRROR>(v);
  v = dpp_add_f32<DPP_MIRROR>(v);
  return (lane_value_f32(v, 0) + lane_value_f32(v, 16)) + (lane_value_f32(v, 32) + lane_value_f32(v, 48));
}

template <int WMAX, bool FAST>
__global__ void __launch_bounds__(64) wide_f32_kernel(const BatchParams P, int JR, int JC, double* out_logdet,
                                                      double* out_quad) {
  using G = WideGeom<WMAX>;
  constexpr int LPR = G::LPR, COLS = G::COLS;
  __shared__ __attribute__((aligned(16))) float ubuf[2][WMAX];
  __shared__ __attribute__((aligned(16))) float pbuf[2][WMAX];
  __shared__ __attribute__((aligned(16))) float wbuf[WMAX];
  const int lane = threadIdx.x, b = blockIdx.x;
  const int row = lane / LPR, seg = lane % LPR;
  const int W = JR + 2 * JC;
  const bool writer = seg == 0;
  RowCoeffs rc{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (row < JR) {
    rc.u0 = P.a_real[(long)b * JR + row]; rc.v0 = 1.0; rc.c = P.c_real[(long)b * JR + row];
  } else if (row < W) {
    const int j = (row - JR) >> 1;
    const double a = P.a_comp[(long)b * JC + j], bb = P.b_comp[(long)b * JC + j];
    if (((row - JR) & 1) == 0) { rc.uc = a; rc.us = bb; rc.vc = 1.0; }
    else                       { rc.uc = -bb; rc.us = a; rc.vs = 1.0; }
    rc.c = P.c_comp[(long)b * JC + j];
    rc.d = P.d_comp[(long)b * JC + j];
  }
  double sum_ar = 0.0, sum_ac = 0.0;
  for (int j = 0; j < JR; ++j) sum_ar += P.a_real[(long)b * JR + j];
  for (int j = 0; j < JC; ++j) sum_ac += P.a_comp[(long)b * JC + j];
  const float a0 = (float)((sum_ar + sum_ac) + P.jitter[b]);
  const double* tp = P.t + b * P.t_stride;
  const double* dp = P.diag + b * P.diag_stride;
  const double* yp = P.y + b * P.y_stride;
  const int N = P.N;
  float S[COLS], f = 0.0f;
#pragma unroll
  for (int c = 0; c < COLS; ++c) S[c] = 0.0f;
  double quad = 0.0;
  LogProduct lp;
  lp.init();
  double tv, dv, yv, tv2;
  {
    const int m = lane;
    tv = m < N ? tp[m] : 0.0; dv = m < N ? dp[m] : 0.0; yv = m < N ? yp[m] : 0.0;
    tv2 = m + 64 < N ? tp[m + 64] : 0.0;
  }
  auto t_at = [&](int k) { return k < 64 ? lane_value(tv, k) : lane_value(tv2, k - 64); };
  double ud, vd, phd;
  row_features<FAST>(rc, t_at(0), 1 < N ? t_at(1) - t_at(0) : 0.0, &ud, &vd, &phd);
  float u = (float)ud, v = (float)vd, phi = (float)phd;
  if (writer) { ubuf[0][row] = u; pbuf[0][row] = phi; }
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int nend = (N - n0 < 64) ? N - n0 : 64;
    for (int k = 0; k < nend; ++k) {
      const int n = n0 + k, cur = n & 1;
      const float diag_n = (float)lane_value(dv, k);
      const float y_n = (float)lane_value(yv, k);
      float u1 = 0.0f, v1 = 0.0f, phi1 = 1.0f;
      if (n + 1 < N) {
        const double t1 = t_at(k + 1);
        const double dx1 = (n + 2 < N) ? t_at(k + 2) - t1 : 0.0;
        row_features<FAST>(rc, t1, dx1, &ud, &vd, &phd);
        u1 = (float)ud; v1 = (float)vd; phi1 = (float)phd;
        if (writer) { ubuf[cur ^ 1][row] = u1; pbuf[cur ^ 1][row] = phi1; }
      }
      float q = 0.0f;
      {
        const float4* uv = reinterpret_cast<const float4*>(&ubuf[cur][seg * COLS]);
#pragma unroll
        for (int c = 0; c < COLS / 4; ++c) {
          const float4 uu = uv[c];
          q = fmaf(S[4 * c], uu.x, q); q = fmaf(S[4 * c + 1], uu.y, q);
          q = fmaf(S[4 * c + 2], uu.z, q); q = fmaf(S[4 * c + 3], uu.w, q);
        }
      }
      if (LPR >= 2) q = dpp_add_f32<DPP_QUAD_XOR1>(q);
      if (LPR >= 4) q = dpp_add_f32<DPP_QUAD_XOR2>(q);
      const float s = row_sum_f32<LPR>(u * q), ub = row_sum_f32<LPR>(u * f);
      const float D = (diag_n + a0) - s;
      const float invD = 1.0f / D;
      const float x = y_n - ub;
      lp.mul((double)D);
      quad += (double)x * (double)x / (double)D;
      const float z = v - q;
      const float w = z * invD;
      if (writer) wbuf[row] = phi * w;
      {
        const float4* pv = reinterpret_cast<const float4*>(&pbuf[cur][seg * COLS]);
        const float4* wv = reinterpret_cast<const float4*>(&wbuf[seg * COLS]);
        const float zr = phi * z;
#pragma unroll
        for (int c = 0; c < COLS / 4; ++c) {
          const float4 pk = pv[c], pw = wv[c];
          S[4 * c] = fmaf(zr, pw.x, (phi * pk.x) * S[4 * c]);
          S[4 * c + 1] = fmaf(zr, pw.y, (phi * pk.y) * S[4 * c + 1]);
          S[4 * c + 2] = fmaf(zr, pw.z, (phi * pk.z) * S[4 * c + 2]);
          S[4 * c + 3] = fmaf(zr, pw.w, (phi * pk.w) * S[4 * c + 3]);
        }
      }
      f = phi * (f + w * x);
      u = u1; v = v1; phi = phi1;
    }
    const int m = n0 + 64 + lane;
    tv = tv2;
    dv = m < N ? dp[m] : 0.0;
    yv = m < N ? yp[m] : 0.0;
    tv2 = m + 64 < N ? tp[m + 64] : 0.0;
  }
  if (lane == 0) {
    out_logdet[b] = lp.log_value();
    out_quad[b] = quad;
  }
}

}  // namespace

int wide_f32_probe_max_width() { return 32; }
void launch_wide_f32_probe(const BatchParams& P, int JR, int JC, double* out_logdet, double* out_quad, hipStream_t s) {
  const int W = JR + 2 * JC;
  const dim3 grid(P.B);
#define CLR_GO(WM)                                                                                              \
  do {                                                                                                          \
    if (P.fast_trig) hipLaunchKernelGGL((wide_f32_kernel<WM, true>), grid, dim3(64), 0, s, P, JR, JC, out_logdet, out_quad);  \
    else hipLaunchKernelGGL((wide_f32_kernel<WM, false>), grid, dim3(64), 0, s, P, JR, JC, out_logdet, out_quad);             \
  } while (0)
  if (W <= 16) CLR_GO(16); else CLR_GO(32);
#undef CLR_GO
}

namespace {
}  // namespace

// the lazy summarize flavours at the padded width 32, built by wide_scan32.hip (this file again, CLR_WIDE_SCAN32_ONLY)
void launch_wide_scan32_lazy(const BatchParams& P, int JR, int JC, bool paired, bool gaps, hipStream_t s);

int wide_max_width() { return 64; }
int wide_scan_max_width() { return 64; }  // the chunk algebra: prefix_coop_kernel<16>, wide_prefix32_kernel, wide_walk64_kernel (wide64_kernels.hip)

// widths 33..64: two waves per (problem, chunk) (wide_scan64x2_kernel); CLR_WIDE64_ONE_WAVE=1 keeps the one-wave kernels (A/B)
static bool wide64_one_wave() { return clr::option("CLR_WIDE64_ONE_WAVE") != nullptr; }
static bool wide_paired(int JR) { return JR == 0 && clr::option("CLR_WIDE_NO_PAIRED") == nullptr; }  // complex terms only: a term's four lanes share its features
template <bool GEN>
static void launch_wide64(const BatchParams& P, int JR, int JC, hipStream_t s) {
  const dim3 grid(P.nchunk, P.B);
  if (!GEN && !wide64_one_wave() && wide_paired(JR)) {
    if (P.fast_trig) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 0, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 0, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
    return;
  }
  if (wide64_one_wave()) {
    if (P.fast_trig) hipLaunchKernelGGL((wide_scan_kernel<64, true, 0, false, GEN>), grid, dim3(64), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan_kernel<64, false, 0, false, GEN>), grid, dim3(64), 0, s, P, JR, JC);
    return;
  }
  if (P.fast_trig) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 0, false, GEN>), grid, dim3(128), 0, s, P, JR, JC);
  else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 0, false, GEN>), grid, dim3(128), 0, s, P, JR, JC);
}

template <int MODE, bool GEN>
static void launch_wide_g(const BatchParams& P, int JR, int JC, hipStream_t s) {
  const int W = JR + 2 * JC + (GEN ? P.J_general : 0);
  const dim3 grid(P.nchunk, P.B);
  if (MODE == 1 && P.split_lazy) {  // summarize with the decay factored out of the state (dense series)
#define CLR_GOL(WM)                                                                                                     \
  do {                                                                                                                  \
    if (P.fast_trig) hipLaunchKernelGGL((wide_scan_kernel<WM, true, 1, true, GEN>), grid, dim3(64), 0, s, P, JR, JC);  \
    else hipLaunchKernelGGL((wide_scan_kernel<WM, false, 1, true, GEN>), grid, dim3(64), 0, s, P, JR, JC);              \
  } while (0)
    // complex terms only: the four lanes of a term share the features' work (wide_scan_body, PAIRED)
    const bool paired = !GEN && wide_paired(JR);
    const bool gaps = !GEN && P.split_lazy == 2 && W > 16;  // (host: not dense everywhere, max c x max dx < 2)
    if (W <= 16) CLR_GOL(16);
    else if (W <= 32 && !GEN && (paired || gaps)) launch_wide_scan32_lazy(P, JR, JC, paired, gaps, s);  // (wide_scan32.hip: its own scheduling strategy)
    else if (W <= 32) CLR_GOL(32);
    else if (!wide64_one_wave() && !GEN && (paired || gaps)) {
      if (P.fast_trig) {
        if (paired && gaps) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, true, false, true, true>), grid, dim3(128), 0, s, P, JR, JC);
        else if (paired) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, true, false, true, false>), grid, dim3(128), 0, s, P, JR, JC);
        else hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, true, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
      } else {
        if (paired && gaps) hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, true, false, true, true>), grid, dim3(128), 0, s, P, JR, JC);
        else if (paired) hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, true, false, true, false>), grid, dim3(128), 0, s, P, JR, JC);
        else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, true, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
      }
    } else if (wide64_one_wave()) {
      if (P.fast_trig) hipLaunchKernelGGL((wide_summarize64_kernel<true, true, GEN>), grid, dim3(64), 0, s, P, JR, JC);
      else hipLaunchKernelGGL((wide_summarize64_kernel<false, true, GEN>), grid, dim3(64), 0, s, P, JR, JC);
    } else if (P.fast_trig) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, true, GEN>), grid, dim3(128), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, true, GEN>), grid, dim3(128), 0, s, P, JR, JC);
#undef CLR_GOL
    return;
  }
  if (MODE == 1 && W > 32) {  // widths 33..64 on a series that is not densely sampled
    if (!GEN && !wide64_one_wave() && wide_paired(JR)) {
      if (P.fast_trig) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
      else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, false, false, true>), grid, dim3(128), 0, s, P, JR, JC);
      return;
    }
    if (wide64_one_wave()) {
      if (P.fast_trig) hipLaunchKernelGGL((wide_summarize64_kernel<true, false, GEN>), grid, dim3(64), 0, s, P, JR, JC);
      else hipLaunchKernelGGL((wide_summarize64_kernel<false, false, GEN>), grid, dim3(64), 0, s, P, JR, JC);
    } else if (P.fast_trig) hipLaunchKernelGGL((wide_scan64x2_kernel<true, 1, false, GEN>), grid, dim3(128), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan64x2_kernel<false, 1, false, GEN>), grid, dim3(128), 0, s, P, JR, JC);
    return;
  }
#define CLR_GO(WM)                                                                                                        \
  do {                                                                                                                    \
    if (P.fast_trig) hipLaunchKernelGGL((wide_scan_kernel<WM, true, MODE, false, GEN>), grid, dim3(64), 0, s, P, JR, JC); \
    else hipLaunchKernelGGL((wide_scan_kernel<WM, false, MODE, false, GEN>), grid, dim3(64), 0, s, P, JR, JC);            \
  } while (0)
  if (W <= 16) CLR_GO(16);
  else if (W <= 32 && !GEN && wide_paired(JR)) {
    if (P.fast_trig) hipLaunchKernelGGL((wide_scan_kernel<32, true, MODE, false, false, true>), grid, dim3(64), 0, s, P, JR, JC);
    else hipLaunchKernelGGL((wide_scan_kernel<32, false, MODE, false, false, true>), grid, dim3(64), 0, s, P, JR, JC);
  } else if (W <= 32 || MODE == 1) CLR_GO(32);
  else launch_wide64<GEN>(P, JR, JC, s);
#undef CLR_GO
}

template <int MODE>
static void launch_wide(const BatchParams& P, int JR, int JC, hipStream_t s) {
  if (P.J_general > 0) launch_wide_g<MODE, true>(P, JR, JC, s);
  else launch_wide_g<MODE, false>(P, JR, JC, s);
}

// ---------------------------------------------------------------------------
// prefix at the padded width 32 on the matrix cores.  One wave per problem walks its chunks; per chunk
// (advance part of chunk_update, clr_core.h):
//     M^T = I + P Jm ;  [M^T | P] --Gauss-Jordan, partial pivoting--> [I | G] ;  Gs = (G + G^T) / 2
//     h = f + P eta ; g = h - G (Jm h) ;  P' = C + (A Gs) A^T ;  f' = A g + b
// All matrices live in LDS (row stride 33 doubles).  The three 32 x 32 x 32 products -- the one dense
// contraction of the whole path -- run as v_mfma_f64_16x16x4_f64: per product 4
// output tiles x 8 k-steps = 32 MFMA instructions, operands read from LDS in the instruction's lane
// layout (A: lane l holds X[m = l & 15][k = l >> 4]; B: Y[k = l >> 4][n = l & 15]; C/D: 4 doubles,
// row (l >> 4) + 4 r, column l & 15).  The elimination works on the 32 x 64 tableau in LDS, lane =
// column: per pivot every lane reads the 32 multipliers (broadcast reads) and updates its own column.
// prefix_coop_kernel<32, 32> did the same algebra with one column per lane in REGISTERS: 512 registers,
// 1 KB of scratch, a ds_bpermute per broadcast -- 300 us per chunk, 97 % of it moving data between lanes.
// ---------------------------------------------------------------------------
struct Prefix32Lds {
  static constexpr int J = 32, LD = 33, LT = 66;
  double P[J * LD], Jf[J * LD], A[J * LD], X[J * LD], C[J * LD], T[J * LT];
  double f[J], h[J], v[J], g[J], eta[J], b[J];
};

// Z = X Y (+ Z0) for 32 x 32 matrices in LDS; YT: Y is given transposed (Y[k][n] = Yt[n][k]);
// IDENT: add the identity; Z0 (may alias nothing written here) initialises the accumulators
template <bool YT, bool IDENT>
__device__ __forceinline__ void mfma_32x32x32(const double* X, int ldx, const double* Y, int ldy, const double* Z0,
                                               int ldz0, double* Z, int ldz, int lane) {
  const int lm = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      mfma_acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lm;
        double init = Z0 ? Z0[row * ldz0 + col] : 0.0;
        if (IDENT && row == col) init += 1.0;
        acc[r] = init;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k = 4 * ks + lk;
        const double a = X[(16 * ti + lm) * ldx + k];
        const double bb = YT ? Y[(16 * tj + lm) * ldy + k] : Y[k * ldy + 16 * tj + lm];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Z[(16 * ti + lk + 4 * r) * ldz + 16 * tj + lm] = acc[r];
    }
  }
}

// one wave per workgroup: its LDS operations execute in program order, so phases that exchange data
// between LANES through LDS only need the compiler not to reorder them and the reads to have landed
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Round 4: FOUR waves per problem.  The walk over the chunks stays sequential, but inside a chunk the element loads are
// strided over 256 threads, every wave computes ONE of the four 16 x 16 output tiles of a product (8 MFMA instead of 32), and
// the elimination is split by rows: wave w owns rows 8 w .. 8 w + 7 of the tableau (lane = column as before), so a pivot
// costs 8 instead of 32 multiplier reads and row updates per lane.  The pivot search and the scaled pivot row belong to
// every wave / the first wave; phases are separated by workgroup barriers.
template <bool YT, bool IDENT>
__device__ __forceinline__ void mfma_tile_32(const double* X, int ldx, const double* Y, int ldy, const double* Z0, int ldz0,
                                             double* Z, int ldz, int lane, int ti, int tj) {
  const int lm = lane & 15, lk = lane >> 4;
  mfma_acc_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lm;
    double init = Z0 ? Z0[row * ldz0 + col] : 0.0;
    if (IDENT && row == col) init += 1.0;
    acc[r] = init;
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int k = 4 * ks + lk;
    const double a = X[(16 * ti + lm) * ldx + k];
    const double bb = YT ? Y[(16 * tj + lm) * ldy + k] : Y[k * ldy + 16 * tj + lm];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Z[(16 * ti + lk + 4 * r) * ldz + 16 * tj + lm] = acc[r];
}

__global__ void __launch_bounds__(256) wide_prefix32_kernel(const BatchParams P_) {
  constexpr int J = 32, SZ = J * (J + 1) / 2, ELEM = J * J + J + SZ + J + SZ, START = SZ + J;
  constexpr int LD = Prefix32Lds::LD, LT = Prefix32Lds::LT, NT = 256;
  __shared__ Prefix32Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, prob = blockIdx.x;
  const int ti = wave >> 1, tj = wave & 1;  // this wave's output tile of the 32 x 32 products
  if (tid == 0) P_.need_exact[prob] = 0;  // raised by wide_correct_kernel / decide_kernel
  for (int idx = tid; idx < J * J; idx += NT) L.P[(idx / J) * LD + idx % J] = 0.0;
  if (tid < J) L.f[tid] = 0.0;
  __syncthreads();
  for (int c = 0; c + 1 < P_.nchunk; ++c) {
    const double* E = P_.elems + ((long)prob * P_.nchunk + c) * ELEM;
    const double* Eb = E + J * J;
    const double* EC = Eb + J;
    const double* Eeta = EC + SZ;
    const double* EJm = Eeta + J;
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      L.A[i * LD + j] = E[idx];
      L.C[i * LD + j] = EC[sym(i, j)];
      L.Jf[i * LD + j] = EJm[sym(i, j)];
      L.T[i * LT + J + j] = L.P[i * LD + j];  // right half of the tableau: P
    }
    if (tid < J) { L.b[tid] = Eb[tid]; L.eta[tid] = Eeta[tid]; }
    __syncthreads();
    // M^T = I + P Jm -> left half of the tableau
    mfma_tile_32<false, true>(L.P, LD, L.Jf, LD, nullptr, 0, L.T, LT, lane, ti, tj);
    // h = f + P eta
    if (tid < J) {
      double acc = L.f[tid];
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc += L.P[tid * LD + j] * L.eta[j];
      L.h[tid] = acc;
    }
    __syncthreads();
    if (tid < J) {  // v = Jm h
      double acc = 0.0;
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc += L.Jf[tid * LD + j] * L.h[j];
      L.v[tid] = acc;
    }
    // Gauss-Jordan with partial pivoting, lane = column of [M^T | P], wave = block of 8 rows
    for (int col = 0; col < J; ++col) {
      int piv = col;
      double best = -1.0;
      for (int i = col; i < J; ++i) {  // (every lane scans the pivot column: broadcast reads)
        const double cand = fabs(L.T[i * LT + col]);
        if (cand > best) { best = cand; piv = i; }
      }
      const double top = L.T[piv * LT + lane], old = L.T[col * LT + lane];
      const double pv = L.T[piv * LT + col];
      __syncthreads();  // (every wave has read the column and the two rows)
      const double t = top * (1.0 / pv);
      if (wave == 0) {
        L.T[piv * LT + lane] = old;   // row swap (a no-op when piv == col) ...
        wave_lds_fence();
        L.T[col * LT + lane] = t;     // ... and the scaled pivot row
      }
      __syncthreads();
      double m[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) m[r] = L.T[(8 * wave + r) * LT + col];
      wave_lds_fence();  // (a wave's own rows: no other wave writes them)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int i = 8 * wave + r;
        if (i != col) L.T[i * LT + lane] = fma(-m[r], t, L.T[i * LT + lane]);
      }
      __syncthreads();
    }
    // Gs = (G + G^T) / 2 -> Jf (Jm is no longer needed); g = h - G v
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      L.Jf[i * LD + j] = 0.5 * (L.T[i * LT + J + j] + L.T[j * LT + J + i]);
    }
    if (tid < J) {
      double acc = L.h[tid];
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc -= L.T[tid * LT + J + j] * L.v[j];
      L.g[tid] = acc;
    }
    __syncthreads();
    // X = A Gs ; P' = C + X A^T ; f' = A g + b
    mfma_tile_32<false, false>(L.A, LD, L.Jf, LD, nullptr, 0, L.X, LD, lane, ti, tj);
    if (tid < J) {
      double acc = L.b[tid];
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc += L.A[tid * LD + j] * L.g[j];
      L.f[tid] = acc;
    }
    __syncthreads();
    mfma_tile_32<true, false>(L.X, LD, L.A, LD, L.C, LD, L.P, LD, lane, ti, tj);
    __syncthreads();
    // start state of chunk c + 1: packed upper triangle (mirrored into the lower one for the next round) | f
    double* o = P_.starts + ((long)prob * P_.nchunk + c + 1) * START;
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (i <= j) o[tri(i, j)] = L.P[i * LD + j];
    }
    if (tid < J) o[SZ + tid] = L.f[tid];
    __syncthreads();
    for (int idx = tid; idx < J * J; idx += NT) {
      const int i = idx / J, j = idx % J;
      if (i > j) L.P[i * LD + j] = L.P[j * LD + i];
    }
    __syncthreads();
  }
}

void launch_wide_prefix(const BatchParams& P, int width_padded, hipStream_t s) {
  if (P.nchunk < 2) return;
  if (P.scan_ws && P.coop_prefix == 2) { launch_wide_prefix_scan(P, width_padded, s); return; }  // (mode 1: the walk)
  if (width_padded <= 16)
    hipLaunchKernelGGL((prefix_coop_kernel<16, 16>), dim3((P.B + 1) / 2), dim3(64), 0, s, P);
  else if (P.coop_prefix)
    hipLaunchKernelGGL(wide_prefix32_kernel, dim3(P.B), dim3(256), 0, s, P);
  else  // (clr_batch_set_prefix_mode(h, 0): the register-resident kernel of round 1, kept for A/B)
    hipLaunchKernelGGL((prefix_coop_kernel<32, 32>), dim3(P.B), dim3(64), 0, s, P);
}

// the scan's correct phase (determinant lemma + Woodbury + certificate) at the padded width
void launch_wide_correct(const BatchParams& P, int width_padded, hipStream_t s) {
  if (P.nchunk < 2) return;
  const dim3 grid((unsigned)((long)P.B * P.nchunk));
  if (width_padded <= 16) hipLaunchKernelGGL((wide_correct_kernel<16>), grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((wide_correct_kernel<32>), grid, dim3(256), 0, s, P);
  launch_wide_decide(P, width_padded, s);
}

// (the corrections' error estimate is J eps / mu per chunk: J = the padded width the correct kernel worked at)
void launch_wide_decide(const BatchParams& P, int width_padded, hipStream_t s) {
  const dim3 block(P.nchunk > 256 ? 256 : 64);
  if (width_padded <= 16) hipLaunchKernelGGL((decide_kernel<16>), dim3(P.B), block, 0, s, P);
  else if (width_padded <= 32) hipLaunchKernelGGL((decide_kernel<32>), dim3(P.B), block, 0, s, P);
  else hipLaunchKernelGGL((decide_kernel<64>), dim3(P.B), block, 0, s, P);
}

// after the chunked replay: a replayed problem whose chunks did not meet the scanned start states
// goes to the sequential sweep (level 2)
__global__ void __launch_bounds__(64) wide_check_replay_kernel(const BatchParams P) {
  // one wave per problem, lanes striding over its chunks (round 4: hundreds of chunks per problem -- a lone thread's
  // dependent loads took 43 us for 390)
  const int b = blockIdx.x, lane = threadIdx.x;
  if (!P.cond) return;
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
  // (level 3: the outputs of a second replay decide -- BatchParams::head_check)
  if (lane == 0 && !(r <= P.cert_resid)) P.need_exact[b] = (P.head_cap > 0.0 && P.ends && P.ends_alt && P.wide_materialize && r <= P.head_cap) ? 3 : 2;
}
// after the output check's pass: level 3 -> 1 (the two replays wrote the same factor to head_tol) or 2 (sequential)
__global__ void __launch_bounds__(64) wide_head_decide_kernel(const BatchParams P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (P.need_exact[b] != 3) return;
  double r = 0.0;
  for (int c = lane; c < P.nchunk; c += 64) {
    const double rc = c == 0 ? 0.0 : P.cond[((long)b * P.nchunk + c) * 3 + 2];   // (chunk 0 starts from the exact zero state: its record is the end-state residual still)
    r = (rc != rc) ? INFINITY : fmax(r, rc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) r = fmax(r, __shfl_xor(r, off, 64));
  if (lane == 0 && (r <= P.head_tol || P.head_check == 2)) P.need_exact[b] = (r <= P.head_tol) ? 1 : 2;   // (head_check == 2: the last attempt)
}
void launch_wide_head_decide(const BatchParams& P, hipStream_t s) {
  hipLaunchKernelGGL(wide_head_decide_kernel, dim3(P.B), dim3(64), 0, s, P);
}
void launch_wide_check_replay(const BatchParams& P, hipStream_t s) {
  hipLaunchKernelGGL(wide_check_replay_kernel, dim3(P.B), dim3(64), 0, s, P);
}

// one chunk: the whole recurrence, results written directly; several chunks: the replay phase
void launch_wide_loglike(const BatchParams& P, int JR, int JC, hipStream_t s) { launch_wide<0>(P, JR, JC, s); }
// the chunks' transfer elements (widths 9..32 only)
void launch_wide_summarize(const BatchParams& P, int JR, int JC, hipStream_t s) { launch_wide<1>(P, JR, JC, s); }

}  // namespace clr
#endif  // CLR_WIDE_SCAN32_ONLY
