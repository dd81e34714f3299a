# -*- coding: utf-8 -*-
"""Dense O(N^3) oracle: builds the covariance matrix explicitly and uses
LAPACK (fp64) or mpmath (extended precision) on it.

TEST INFRASTRUCTURE ONLY (see oracle/celerite_ref.h).  This is the comparator
the reference's own tests use (tests/test_celerite.py:59-64, 116-129, 365-370:
``get_kernel_value`` + ``np.linalg.slogdet`` / ``np.linalg.solve``), restated
with NumPy so it runs without the reference.
"""
import numpy as np


def kernel_value(a_real, c_real, a_comp, b_comp, c_comp, d_comp, tau):
    """k(tau) of cpp/include/celerite/utils.h:106-132, vectorised over tau."""
    tau = np.abs(np.asarray(tau, dtype=np.float64))
    k = np.zeros_like(tau)
    for a, c in zip(np.atleast_1d(a_real), np.atleast_1d(c_real)):
        k = k + a * np.exp(-c * tau)
    for a, b, c, d in zip(*(np.atleast_1d(v) for v in (a_comp, b_comp, c_comp, d_comp))):
        k = k + np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))
    return k


def dense_matrix(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V, t, diag):
    """K = k(|t_i - t_j|) + diag(diag + jitter [+ A]) [+ tril(U^T V) + triu(V^T U)].

    Same construction as tests/test_celerite.py:116-124.
    """
    t = np.asarray(t, dtype=np.float64)
    K = kernel_value(a_real, c_real, a_comp, b_comp, c_comp, d_comp, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += np.asarray(diag, dtype=np.float64) + jitter
    A = np.asarray(A, dtype=np.float64)
    if A.size:
        U = np.asarray(U, dtype=np.float64)
        V = np.asarray(V, dtype=np.float64)
        K[np.diag_indices_from(K)] += A
        K += np.tril(np.dot(U.T, V), -1) + np.triu(np.dot(V.T, U), 1)
    return K


def dense_logdet(K):
    return np.linalg.slogdet(K)[1]


def dense_log_likelihood(K, y):
    y = np.asarray(y, dtype=np.float64)
    quad = float(np.dot(y, np.linalg.solve(K, y)))
    ld = float(np.linalg.slogdet(K)[1])
    return -0.5 * (quad + ld + len(y) * np.log(2 * np.pi)), ld, quad


def mp_logdet_quad(K, y, dps=40):
    """Extended-precision log det K and y^T K^-1 y by dense LDL^T in mpmath.

    K's entries are the fp64 values handed in (so this measures the error of the
    *factorisation and solve*, not of building K).  O(N^3) in Python: N <~ 300.
    """
    import mpmath as mp

    mp.mp.dps = dps
    n = K.shape[0]
    L = [[mp.mpf(0)] * n for _ in range(n)]
    D = [mp.mpf(0)] * n
    Kmp = [[mp.mpf(float(K[i, j])) for j in range(n)] for i in range(n)]
    for j in range(n):
        s = Kmp[j][j]
        for k in range(j):
            s -= L[j][k] * L[j][k] * D[k]
        D[j] = s
        L[j][j] = mp.mpf(1)
        for i in range(j + 1, n):
            s = Kmp[i][j]
            for k in range(j):
                s -= L[i][k] * L[j][k] * D[k]
            L[i][j] = s / D[j]
    logdet = sum(mp.log(d) for d in D)
    z = [mp.mpf(float(v)) for v in y]
    for i in range(n):
        for k in range(i):
            z[i] -= L[i][k] * z[k]
    quad = sum(z[i] * z[i] / D[i] for i in range(n))
    return float(logdet), float(quad)


def mp_exact_logdet_quad(a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, dps=60):
    """log det K and y^T K^-1 y with K BUILT and factorised in `dps`-digit arithmetic from
    the fp64 inputs (kernel of celerite/terms.py:59-65 / utils.h:106-132): the exact answer
    both the reference's recurrence and the scan are approximations of.  N <~ 100."""
    import mpmath as mp

    mp.mp.dps = dps
    n = len(t)
    tm = [mp.mpf(float(v)) for v in t]
    ar = [mp.mpf(float(v)) for v in np.atleast_1d(a_real)]
    cr = [mp.mpf(float(v)) for v in np.atleast_1d(c_real)]
    ac = [mp.mpf(float(v)) for v in np.atleast_1d(a_comp)]
    bc = [mp.mpf(float(v)) for v in np.atleast_1d(b_comp)]
    cc = [mp.mpf(float(v)) for v in np.atleast_1d(c_comp)]
    dc = [mp.mpf(float(v)) for v in np.atleast_1d(d_comp)]

    def kern(tau):
        s = mp.mpf(0)
        for a, c in zip(ar, cr):
            s += a * mp.exp(-c * tau)
        for a, b, c, d in zip(ac, bc, cc, dc):
            s += mp.exp(-c * tau) * (a * mp.cos(d * tau) + b * mp.sin(d * tau))
        return s

    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(i + 1):
            K[i, j] = K[j, i] = kern(abs(tm[i] - tm[j]))
        K[i, i] += mp.mpf(float(diag[i]))
    L = mp.cholesky(K)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    yv = mp.matrix([mp.mpf(float(v)) for v in y])
    z = mp.lu_solve(K, yv)
    quad = sum(yv[i] * z[i] for i in range(n))
    return float(logdet), float(quad)
