// celerite_amd/csrc/clr_prefix_kernels.h
//
// Multi-level prefix of the chunked scan (widths J <= 8): the "Blelloch" part of the path.
//
// The plain prefix (prefix_coop_kernel, clr_batch_kernels.h) walks a problem's nchunk
// elements one after the other: nchunk dependent Gauss-Jordan eliminations of ~2.8 us each
// on a lone wave, with three quarters of the chip idle.  The elements compose in closed
// form (clr_core.h: compose_elements), so the walk is cut into groups:
//
//   up    group_compose_kernel   every group of g consecutive level-l elements is composed
//                                into ONE level-(l+1) element (g - 1 compositions deep, all
//                                groups of all problems side by side);
//   top   seg_advance_kernel     the few top-level elements are walked from the zero state;
//   down  seg_advance_kernel     every group walks its own g elements from the start state
//                                the level above found for it (g - 1 advances deep).
//
// Dependent chain: sum over levels of ~2.2 (g - 1) + n_top advances instead of nchunk
// (clr_core.h: plan_prefix); 64 chunks in groups of 4: 15 + 3.6 + 3 against 63.
//
// Lane mapping (both kernels): a 16-lane DPP row runs one Gauss-Jordan on [M^T | R]:
// lanes 0..7 hold the columns of M^T = I + P Jm, lanes 8..15 the columns of the right-hand
// sides R; pivot row and multipliers travel by DPP row broadcasts (clr_batch_kernels.h:
// row_bcast).  seg_advance: one row per segment, R = P (the running state), four segments
// per wave -- prefix_coop_kernel's arithmetic with a start state and a segment view.
// group_compose: TWO rows per group.  With the running composition e1 = (A1, b1, C1, eta1,
// Jm1) and the next element e2, (C12, b12) is e2 advancing the state (C1, b1), so the
// "state row" does exactly an advance with R = C1; the "rider row" eliminates the same
// M^T (duplicated, so that its broadcasts stay inside the row) with R = A1 and gets
// X1 = (I + C1 Jm2)^-1 A1, from which
//   A12 = A2 X1 ,  Jm12 = Jm1 + A1^T Jm2 X1 ,  eta12 = eta1 + X1^T (eta2 - Jm2 b1) .
// Both rows share one instruction stream; a composition costs one advance plus one J^3/16
// product.  Products with a matrix every lane needs (A2, Jm2, A1) read it from LDS
// (broadcast reads), the per-lane operand stays in registers.
// (included from the middle of clr_batch_kernels.h: BatchParams and the DPP helpers are defined above)
#pragma once

namespace clr {

struct SegParams {
  const double* elems;          // level-l elements      [B][n][ELEM]
  double* starts;               // level-l start states  [B][n][START]   (seg_advance: out)
  const double* parent_starts;  // level-(l+1) start states [B][np][START]; null: the zero state
  double* parents;              // level-(l+1) elements  [B][np][ELEM]    (group_compose: out)
  int B, n, g, np;              // np = ceil(n / g) segments per problem
  int* need_exact;              // [B], cleared when non-null (raised afterwards by correct_kernel)
};

// One Gauss-Jordan elimination with partial pivoting on the 16 columns of a DPP row
// (lanes 0..7: M^T, lanes 8..15: right-hand sides), column per lane.
template <int J>
__device__ __forceinline__ void gauss_jordan_row16(double (&T)[J]) {
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
    piv = row_bcast_int(piv, c0);  // the decision of the pivot column's lane
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
    for (int i = 0; i < J; ++i) m[i] = row_bcast(T[i], c0);
    const double t = T[c0] * recip_fast(m[c0]);  // (a zero / non-finite pivot gives NaN: the problem is then flagged downstream)
#pragma unroll
    for (int i = 0; i < J; ++i) T[i] = (i == c0) ? t : (T[i] - m[i] * t);
  }
}

// ---------------------------------------------------------------------------
// seg_advance: segment s = (problem b, group k) walks elements [k g, k g + len) of its
// problem from the start state parent_starts[b][k] (zero when null) and writes the start
// state of every element of the segment.
// ---------------------------------------------------------------------------
template <int J>
__global__ void __launch_bounds__(64) seg_advance_kernel(const SegParams S) {
  constexpr int HALF = 8, GROUP = 16, NG = 4;
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  constexpr int START = SZ + J;
  __shared__ double pbuf[NG][HALF][HALF];      // pbuf[g][j][i] = P[i][j] (column j contiguous)
  __shared__ double xbuf[NG][HALF][HALF + 1];  // transpose buffer, padded
  constexpr int ESTRIDE = ((ELEM + 15) / 16) * 16 + 2;
  constexpr int EPER = (ELEM + GROUP - 1) / GROUP;
  __shared__ double ebuf[2][NG][ESTRIDE];
  const int lane = threadIdx.x, g = lane / GROUP, l = lane % GROUP;
  const bool rhs = l >= HALF;
  const int col = l % HALF;
  const bool cv = col < J;
  const int cc = cv ? col : J - 1;
  const long nseg = (long)S.B * S.np;
  long seg = (long)blockIdx.x * NG + g;
  const bool active = seg < nseg;
  if (!active) seg = nseg - 1;
  const long sb = seg / S.np;
  const int sk = (int)(seg % S.np);
  const int first = sk * S.g;
  const int len = min(S.g, S.n - first);
  const bool writer = rhs && cv && active;
  const long ebase = sb * S.n;  // element / start index of the problem's first element

  if (S.need_exact && l == 0 && active && sk == 0) S.need_exact[sb] = 0;

  double Pc[J];  // column `col` of the running P (rhs lanes)
  double fj = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) Pc[i] = 0.0;
  if (S.parent_starts && rhs) {
    const double* ps = S.parent_starts + seg * START;
#pragma unroll
    for (int i = 0; i < J; ++i) Pc[i] = cv ? ps[sym(i, cc)] : 0.0;
    fj = cv ? ps[SZ + cc] : 0.0;
  }
  if (rhs) {
#pragma unroll
    for (int i = 0; i < HALF; ++i) pbuf[g][col][i] = (i < J) ? Pc[i < J ? i : 0] : 0.0;
  }
  if (writer) {  // the segment's first element starts from the incoming state
    double* o = S.starts + (ebase + first) * START;
#pragma unroll
    for (int k = 0; k < J; ++k)
      if (k <= col) o[tri(k, cc)] = Pc[k];
    o[SZ + cc] = fj;
  }
  {
    const double* E0 = S.elems + (ebase + first) * ELEM;
#pragma unroll
    for (int m = 0; m < EPER; ++m)
      if (l + GROUP * m < ELEM) ebuf[0][g][l + GROUP * m] = E0[l + GROUP * m];
  }
  __syncthreads();

  // (wave-uniform trip count: a ragged last group idles through its missing steps)
  for (int c = 0; c + 1 < S.g; ++c) {
    const bool valid = c + 1 < len;
    double nx[EPER];  // this lane's share of the next element, in flight during the whole iteration
    const bool more = c + 2 < S.g;
    if (more) {
      const double* En = S.elems + (ebase + first + min(c + 1, len - 1)) * ELEM;
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
    double T[J];  // column of [M^T | P]
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double acc = (i == col) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc += pbuf[g][j][i] * jc[j];
      T[i] = rhs ? Pc[i] : acc;
    }
    double hj = fj;  // h = f + P eta (component `col` in rhs lane `col`), v = Jm h
#pragma unroll
    for (int i = 0; i < J; ++i) hj += Pc[i] * et[i];
    double vj = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) vj += jc[i] * row_bcast(hj, HALF + i);

    gauss_jordan_row16<J>(T);
    // rhs lanes: T = G[:, col] = G[col, :]
    double gj = hj;  // g = h - G v
#pragma unroll
    for (int i = 0; i < J; ++i) gj -= T[i] * row_bcast(vj, HALF + i);
    double fn = cv ? bv[cc] : 0.0;  // f' = A g + b
#pragma unroll
    for (int i = 0; i < J; ++i) fn += (cv ? A[cc * J + i] : 0.0) * row_bcast(gj, HALF + i);

    double Xr[J];  // X = G A^T: lane `col` computes row `col`; transposed through LDS
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
    if (rhs && valid) {
#pragma unroll
      for (int i = 0; i < J; ++i) {
        Pc[i] = cv ? Pn[i] : 0.0;
        pbuf[g][col][i] = Pc[i];
      }
      fj = fn;
    }
    if (writer && valid) {
      double* o = S.starts + (ebase + first + c + 1) * START;
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
// group_compose: segment s = (problem b, group k) composes elements [k g, k g + len) of
// its problem, in order, into parents[b][k].  32 lanes per segment: a state row and a
// rider row (header comment); two segments per wave.
// ---------------------------------------------------------------------------
template <int J>
__global__ void __launch_bounds__(64) group_compose_kernel(const SegParams S) {
  constexpr int HALF = 8, ROW = 16, GROUP = 32, NG = 2;
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  // The running composition (A1, b1, C1, eta1, Jm1) lives in LDS, double-buffered: an iteration reads buffer p and
  // writes buffer p ^ 1, so the registers hold only an iteration's temporaries (with the element in registers the
  // kernel needed 308 of them at width 8: one wave per SIMD and four rounds of waves at the headline shape).
  __shared__ double pbuf[2][NG][HALF][HALF];   // C1: [j][i] = C1[i][j] (column j contiguous)
  __shared__ double abuf[2][NG][J * J + 2];    // A1^T row-major: [k * J + a] = A1[a][k] (column k of A1 contiguous)
  __shared__ double jbuf[2][NG][HALF][HALF];   // Jm1: [j][i] = Jm1[i][j]
  __shared__ double bbuf[2][NG][HALF];         // b1
  __shared__ double hbuf[2][NG][HALF];         // eta1
  __shared__ double xbuf[2 * NG][HALF][HALF + 1];  // per row: the exchange of product 1
  constexpr int ESTRIDE = ((ELEM + 15) / 16) * 16 + 2;
  constexpr int EPER = (ELEM + GROUP - 1) / GROUP;
  __shared__ double ebuf[2][NG][ESTRIDE];
  const int lane = threadIdx.x, g = lane / GROUP, lg = lane % GROUP, row = lane / ROW, l = lane % ROW;
  const bool rider = (lg / ROW) != 0;
  const bool rhs = l >= HALF;
  const int col = l % HALF;
  const bool cv = col < J;
  const int cc = cv ? col : J - 1;
  const long nseg = (long)S.B * S.np;
  long seg = (long)blockIdx.x * NG + g;
  const bool active = seg < nseg;
  if (!active) seg = nseg - 1;
  const long sb = seg / S.np;
  const int sk = (int)(seg % S.np);
  const int first = sk * S.g;
  const int len = min(S.g, S.n - first);
  const long ebase = sb * S.n;

  {
    const double* E0 = S.elems + (ebase + first) * ELEM;
#pragma unroll
    for (int m = 0; m < EPER; ++m)
      if (lg + GROUP * m < ELEM) ebuf[0][g][lg + GROUP * m] = E0[lg + GROUP * m];
  }
  __syncthreads();
  if (rhs) {  // the group's first element is the initial composition: buffer 0
    const double* E = ebuf[0][g];
    const double* A = E;
    const double* bv = E + J * J;
    const double* C = bv + J;
    const double* eta = C + SZ;
    const double* Jm = eta + J;
    if (!rider) {
#pragma unroll
      for (int i = 0; i < HALF; ++i) pbuf[0][g][col][i] = (cv && i < J) ? C[sym(i < J ? i : 0, cc)] : 0.0;
      bbuf[0][g][col] = cv ? bv[cc] : 0.0;
    } else {
#pragma unroll
      for (int i = 0; i < HALF; ++i) jbuf[0][g][col][i] = (cv && i < J) ? Jm[sym(i < J ? i : 0, cc)] : 0.0;
      hbuf[0][g][col] = cv ? eta[cc] : 0.0;
      if (cv) {
#pragma unroll
        for (int i = 0; i < J; ++i) abuf[0][g][col * J + i] = A[i * J + cc];
      }
    }
  }
  {
    const double* E1 = S.elems + (ebase + first + min(1, len - 1)) * ELEM;
#pragma unroll
    for (int m = 0; m < EPER; ++m)
      if (lg + GROUP * m < ELEM) ebuf[1][g][lg + GROUP * m] = E1[lg + GROUP * m];
  }
  __syncthreads();

  int p = 0;  // buffer holding the running composition
  for (int c = 1; c < S.g; ++c, p ^= 1) {  // wave-uniform trip count; e2 = element first + c
    const bool valid = c < len;
    double nx[EPER];
    const bool more = c + 1 < S.g;
    if (more) {
      const double* En = S.elems + (ebase + first + min(c + 1, len - 1)) * ELEM;
#pragma unroll
      for (int m = 0; m < EPER; ++m) nx[m] = (lg + GROUP * m < ELEM) ? En[lg + GROUP * m] : 0.0;
    }
    const double* E = ebuf[c & 1][g];
    const double* A = E;  // A2, b2, C2, eta2, Jm2
    const double* bv = E + J * J;
    const double* C = bv + J;
    const double* eta = C + SZ;
    const double* Jm = eta + J;

    double jc[J];
#pragma unroll
    for (int i = 0; i < J; ++i) jc[i] = cv ? Jm[sym(i, cc)] : 0.0;
    double T[J];  // column of [I + C1 Jm2 | C1] (state row) / [I + C1 Jm2 | A1] (rider row)
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double acc = (i == col) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc += pbuf[p][g][j][i] * jc[j];
      const double mine = rider ? (cv ? abuf[p][g][cc * J + i] : 0.0) : pbuf[p][g][col][i];
      T[i] = rhs ? mine : acc;
    }
    // state row: h = b1 + C1 eta2, v = Jm2 h;  rider row: v = Jm2 b1, w = eta2 - v
    double hj = bbuf[p][g][col];
    if (!rider) {
#pragma unroll
      for (int i = 0; i < J; ++i) hj += T[i] * eta[i];  // (rhs lanes: T = C1[:, col])
    }
    double vj = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) vj += jc[i] * row_bcast(hj, HALF + i);
    const double zj = rider ? ((cv ? eta[cc] : 0.0) - vj) : vj;

    gauss_jordan_row16<J>(T);
    // rhs lanes: T = X2[:, col] (= G, symmetric) in the state row, X1[:, col] in the rider row
    double acc1 = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) acc1 += T[i] * row_bcast(zj, HALF + i);
    const double gj = hj - acc1;                       // state row: g = h - G v
    const double etan = hbuf[p][g][col] + acc1;        // rider row: eta12[col] = eta1[col] + X1[:, col] . w
    double fn = cv ? bv[cc] : 0.0;                     // state row: b12 = A2 g + b2
#pragma unroll
    for (int i = 0; i < J; ++i) fn += (cv ? A[cc * J + i] : 0.0) * row_bcast(gj, HALF + i);

    // product 1: A2 T = row `col` of G A2^T (state row, exchanged below) / column `col` of A12 (rider row, stored)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < J; ++i) acc += T[i] * A[j * J + i];
      if (rhs) {
        if (rider) { if (cv && valid) abuf[p ^ 1][g][cc * J + j] = acc; }
        else xbuf[row][col][j] = acc;
      }
    }
    // rider row: its own Jm2 X1[:, col], stored transposed so that both rows read xbuf[row][a][col] below
    if (rider) {
#pragma unroll
      for (int k = 0; k < J; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < J; ++a) acc += Jm[sym(k, a)] * T[a];
        if (rhs) xbuf[row][k][col] = acc;
      }
    }
    __syncthreads();
    // product 2: add + L1 xbuf[row][:, col];  state row: C2[:, col] + A2 (...) = C12[:, col];
    // rider row: Jm1[:, col] + A1^T (Jm2 X1[:, col]) = Jm12[:, col]
    const double* L1 = rider ? abuf[p][g] : A;
#pragma unroll
    for (int k = 0; k < J; ++k) {
      double acc = rider ? jbuf[p][g][col][k] : (cv ? C[sym(k, cc)] : 0.0);
#pragma unroll
      for (int a = 0; a < J; ++a) acc += L1[k * J + a] * xbuf[row][a][col];
      if (rhs) {
        if (!valid) acc = rider ? jbuf[p][g][col][k] : pbuf[p][g][col][k];  // (a ragged group idles: carry over)
        if (rider) jbuf[p ^ 1][g][col][k] = cv ? acc : 0.0;
        else pbuf[p ^ 1][g][col][k] = cv ? acc : 0.0;
      }
    }
    if (rhs) {
      if (rider) {
        hbuf[p ^ 1][g][col] = valid ? etan : hbuf[p][g][col];
        if (cv && !valid) {
#pragma unroll
          for (int i = 0; i < J; ++i) abuf[p ^ 1][g][cc * J + i] = abuf[p][g][cc * J + i];
        }
      } else {
        bbuf[p ^ 1][g][col] = valid ? fn : bbuf[p][g][col];
      }
    }
    if (more) {
#pragma unroll
      for (int m = 0; m < EPER; ++m)
        if (lg + GROUP * m < ELEM) ebuf[(c + 1) & 1][g][lg + GROUP * m] = nx[m];
    }
    __syncthreads();
  }

  if (rhs && cv && active) {
    double* o = S.parents + seg * ELEM;
    if (rider) {
#pragma unroll
      for (int i = 0; i < J; ++i) o[i * J + cc] = abuf[p][g][cc * J + i];  // A row-major
      o[J * J + J + SZ + cc] = hbuf[p][g][col];                           // eta
      double* oj = o + J * J + J + SZ + J;
#pragma unroll
      for (int k = 0; k < J; ++k)
        if (k <= col) oj[tri(k, cc)] = jbuf[p][g][col][k];
    } else {
      o[J * J + cc] = bbuf[p][g][col];  // b
      double* oc = o + J * J + J;
#pragma unroll
      for (int k = 0; k < J; ++k)
        if (k <= col) oc[tri(k, cc)] = pbuf[p][g][col][k];
    }
  }
}

// Single-lane composition of the same groups (clr_core.h: compose_elements, the host-checked form):
// the on-device cross-check of group_compose_kernel (clr_batch_debug_compose_check).
template <int J>
__global__ void __launch_bounds__(64) group_compose_reference_kernel(const SegParams S) {
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  const long seg = (long)blockIdx.x * 64 + threadIdx.x;
  if (seg >= (long)S.B * S.np) return;
  const long sb = seg / S.np;
  const int sk = (int)(seg % S.np);
  const int first = sk * S.g;
  const int len = min(S.g, S.n - first);
  double e[ELEM];
  const double* src = S.elems + (sb * S.n + first) * ELEM;
  for (int i = 0; i < ELEM; ++i) e[i] = src[i];
  for (int c = 1; c < len; ++c) compose_elements<J>(e, src + (long)c * ELEM, e);
  double* o = S.parents + seg * ELEM;
  for (int i = 0; i < ELEM; ++i) o[i] = e[i];
}

// The whole multi-level prefix of a batch: level buffers lvl_elems / lvl_starts hold levels 1..plan.levels
// back to back ([B][n[l]] elements / start states each).
template <int J>
void launch_multilevel_prefix(const BatchParams& P, hipStream_t s) {
  constexpr int SZ = J * (J + 1) / 2;
  constexpr int ELEM = J * J + J + SZ + J + SZ;
  constexpr int START = SZ + J;
  const PrefixPlan& plan = P.plan;
  const double* elems[4];
  double* starts[4];
  elems[0] = P.elems;
  starts[0] = P.starts;
  {
    double* e = P.lvl_elems;
    double* st = P.lvl_starts;
    for (int l = 1; l <= plan.levels; ++l) {
      elems[l] = e;
      starts[l] = st;
      e += (size_t)P.B * plan.n[l] * ELEM;
      st += (size_t)P.B * plan.n[l] * START;
    }
  }
  for (int l = 0; l < plan.levels; ++l) {  // up
    SegParams S{elems[l], nullptr, nullptr, const_cast<double*>(elems[l + 1]), P.B, plan.n[l], plan.g[l], plan.n[l + 1],
                nullptr};
    const long nseg = (long)P.B * S.np;
    hipLaunchKernelGGL((group_compose_kernel<J>), dim3((unsigned)((nseg + 1) / 2)), dim3(64), 0, s, S);
  }
  {  // top: one segment per problem
    const int l = plan.levels;
    SegParams S{elems[l], starts[l], nullptr, nullptr, P.B, plan.n[l], plan.n[l], 1, P.need_exact};
    hipLaunchKernelGGL((seg_advance_kernel<J>), dim3((unsigned)((P.B + 3) / 4)), dim3(64), 0, s, S);
  }
  for (int l = plan.levels - 1; l >= 0; --l) {  // down
    SegParams S{elems[l], starts[l], starts[l + 1], nullptr, P.B, plan.n[l], plan.g[l], plan.n[l + 1], nullptr};
    const long nseg = (long)P.B * S.np;
    hipLaunchKernelGGL((seg_advance_kernel<J>), dim3((unsigned)((nseg + 3) / 4)), dim3(64), 0, s, S);
  }
}

// doubles of level workspace a plan needs per problem: elements, start states
inline void multilevel_workspace(const PrefixPlan& plan, int J, size_t* elem_doubles, size_t* start_doubles) {
  const size_t SZ = (size_t)J * (J + 1) / 2, ELEM = (size_t)J * J + J + SZ + J + SZ, START = SZ + J;
  size_t ne = 0;
  for (int l = 1; l <= plan.levels; ++l) ne += (size_t)plan.n[l];
  *elem_doubles = ne * ELEM;
  *start_doubles = ne * START;
}

}  // namespace clr
