# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE ONLY -- CPU oracle for ``CholeskySolver.grad_log_likelihood``.

The reference obtains the gradient by instantiating its solver template with
Eigen's forward-mode ``AutoDiffScalar<VectorXd>`` (celerite/solver.cpp:347-463) and
running ``compute`` (cpp/include/celerite/solver/cholesky.h:41-210) and ``dot_solve``
(:326-401) on dual numbers.  This module restates exactly that: the same loops, in the
same order, on a small dual-number class (value + a numpy vector of partials).  Pure
Python loops over n: meant for the sizes of the reference's own gradient test
(tests/test_celerite.py:425-481, N = 100) up to a few thousand samples.

Parity status: Eigen and autograd are absent from this image, so the reference's
gradient cannot be executed here; this oracle is pinned (tests/test_oracle.py) by
  * its VALUE against oracle/celerite_ref.c (already pinned to the reference's goldens),
  * central finite differences of that pinned log-likelihood, the criterion the
    reference's own test uses (tests/test_celerite.py:452-481, eps = 1.34e-7).
"""
import math

import numpy as np

DBL_EPSILON = 2.220446049250313e-16


class Dual(object):
    """value + partials, the subset of AutoDiffScalar the solver loops need."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v = float(v)
        self.d = d

    @staticmethod
    def const(v, G):
        return Dual(v, np.zeros(G))

    @staticmethod
    def var(v, G, i):
        d = np.zeros(G)
        d[i] = 1.0
        return Dual(v, d)

    def _lift(self, o):
        return o if isinstance(o, Dual) else Dual(o, np.zeros_like(self.d))

    def __add__(self, o):
        o = self._lift(o)
        return Dual(self.v + o.v, self.d + o.d)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        return Dual(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __neg__(self):
        return Dual(-self.v, -self.d)

    def __mul__(self, o):
        o = self._lift(o)
        return Dual(self.v * o.v, self.d * o.v + self.v * o.d)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._lift(o)
        q = self.v / o.v
        return Dual(q, (self.d - q * o.d) / o.v)

    def __rtruediv__(self, o):
        return self._lift(o) / self


def dexp(x):
    e = math.exp(x.v)
    return Dual(e, e * x.d)


def dlog(x):
    return Dual(math.log(x.v), x.d / x.v)


def dcos(x):
    return Dual(math.cos(x.v), -math.sin(x.v) * x.d)


def dsin(x):
    return Dual(math.sin(x.v), math.cos(x.v) * x.d)


class LinAlgError(Exception):
    """celerite::linalg_exception (cholesky.h:176)."""


def grad_log_likelihood(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V, x, y, diag):
    """(value, gradient) exactly as celerite/solver.cpp:347-463 returns them: the
    gradient has 1 + 2 J_real + 4 J_comp entries -- jitter first (0.0 when the jitter is
    not above DBL_EPSILON, solver.cpp:379-389,419-426), then a_real, c_real, a_comp,
    b_comp, c_comp, d_comp; the value carries the reference's constant
    ``M_PI * log(N)`` (solver.cpp:415), NOT ``N log 2 pi``."""
    a_real, c_real, a_comp, b_comp, c_comp, d_comp = [np.atleast_1d(np.asarray(v, dtype=np.float64))
                                                      for v in (a_real, c_real, a_comp, b_comp, c_comp, d_comp)]
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    diag = np.asarray(diag, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64).reshape(-1, len(x)) if np.size(U) else np.empty((0, len(x)))
    V = np.asarray(V, dtype=np.float64).reshape(-1, len(x)) if np.size(V) else np.empty((0, len(x)))
    J_real, J_comp, J_general = len(a_real), len(a_comp), U.shape[0]
    N = len(x)
    J = J_real + 2 * J_comp + J_general
    has_general = A.size != 0

    G = 2 * J_real + 4 * J_comp                      # solver.cpp:371
    compute_jitter = jitter > DBL_EPSILON             # solver.cpp:379-389
    i0 = 0
    if compute_jitter:
        G += 1
        jit = Dual.var(jitter, G, 0)
        i0 = 1
    else:
        jit = Dual.const(jitter, G)
    ar = [Dual.var(a_real[i], G, i0 + i) for i in range(J_real)]; i0 += J_real          # :392-397
    cr = [Dual.var(c_real[i], G, i0 + i) for i in range(J_real)]; i0 += J_real
    ac = [Dual.var(a_comp[i], G, i0 + i) for i in range(J_comp)]; i0 += J_comp          # :398-406
    bc = [Dual.var(b_comp[i], G, i0 + i) for i in range(J_comp)]; i0 += J_comp
    cc = [Dual.var(c_comp[i], G, i0 + i) for i in range(J_comp)]; i0 += J_comp
    dc = [Dual.var(d_comp[i], G, i0 + i) for i in range(J_comp)]

    zero = Dual.const(0.0, G)

    # ---- compute: cholesky.h:98-210 on duals ------------------------------------
    asum = zero
    for a in ar:
        asum = asum + a
    csum = zero
    for a in ac:
        csum = csum + a
    D = []
    for n in range(N):                                 # :98-99
        dn = ((diag[n] + asum) + csum) + jit
        if has_general:
            dn = dn + A[n]
        D.append(dn)
    if J == 0:                                         # :90-95
        log_det = zero
        for n in range(N):
            log_det = log_det + dlog(D[n])
        quad = zero
        for n in range(N):
            quad = quad + y[n] * (y[n] / D[n])         # :337-338
        return _finish(quad, log_det, N, compute_jitter, G)

    W_prev = [zero] * J
    Dn = D[0]
    value = 1.0 / Dn                                   # :103-117
    t0 = x[0]
    for j in range(J_real):
        W_prev[j] = value
    k = J_real
    for j in range(J_comp):
        d = dc[j] * t0
        W_prev[k] = dcos(d) * value
        W_prev[k + 1] = dsin(d) * value
        k += 2
    for j in range(J_general):
        W_prev[k] = V[j, 0] * value
        k += 1

    S = [[zero] * J for _ in range(J)]                 # S[k][j], k <= j  (:124)
    f = [zero] * J                                     # dot_solve state (:344)
    xm1 = Dual.const(y[0], G)
    result = xm1 * (xm1 / D[0])                        # :347

    for n in range(1, N):                              # :126-179
        t = x[n]
        dx = t - x[n - 1]
        phi = [zero] * J
        u = [zero] * J
        Wn = [zero] * J
        for j in range(J_real):                        # :129-133
            phi[j] = dexp(-cr[j] * dx)
            u[j] = ar[j]
            Wn[j] = Dual.const(1.0, G)
        k = J_real
        for j in range(J_comp):                        # :134-147
            a, b = ac[j], bc[j]
            d = dc[j] * t
            cd, sd = dcos(d), dsin(d)
            value = dexp(-cc[j] * dx)
            phi[k] = value
            phi[k + 1] = value
            u[k] = a * cd + b * sd
            u[k + 1] = a * sd - b * cd
            Wn[k] = cd
            Wn[k + 1] = sd
            k += 2
        for j in range(J_general):                     # :148-152
            phi[k] = Dual.const(1.0, G)
            u[k] = Dual.const(U[j, n], G)
            Wn[k] = Dual.const(V[j, n], G)
            k += 1

        for j in range(J):                             # :154-160
            xj = Dn * W_prev[j]
            for kk in range(j + 1):
                S[kk][j] = phi[j] * (phi[kk] * (S[kk][j] + xj * W_prev[kk]))

        Dn = D[n]                                      # :162-175
        for j in range(J):
            uj = u[j]
            xj = Wn[j]
            for kk in range(j):
                tmp = u[kk] * S[kk][j]
                Dn = Dn - 2.0 * (uj * tmp)
                xj = xj - tmp
                Wn[kk] = Wn[kk] - uj * S[kk][j]
            tmp = uj * S[j][j]
            Dn = Dn - uj * tmp
            Wn[j] = xj - tmp
        if Dn.v < 0:                                   # :176
            raise LinAlgError("failed to factorize or solve matrix")
        D[n] = Dn
        for j in range(J):                             # :178
            Wn[j] = Wn[j] / Dn

        # dot_solve, cholesky.h:348-357, fused into the same sweep (it only looks back)
        xv = Dual.const(y[n], G)
        for j in range(J):
            value = phi[j] * (f[j] + W_prev[j] * xm1)
            f[j] = value
            xv = xv - u[j] * value
        xm1 = xv
        result = result + xv * xv / D[n]
        W_prev = Wn

    log_det = zero                                     # :208
    for n in range(N):
        log_det = log_det + dlog(D[n])
    return _finish(result, log_det, N, compute_jitter, G)


def _finish(quad, log_det, N, compute_jitter, G):
    ll = -0.5 * (quad + log_det + math.pi * math.log(N))      # solver.cpp:415
    if compute_jitter:                                         # :419-426
        g = ll.d.copy()
    else:
        g = np.concatenate([[0.0], ll.d])
    return ll.v, g
