# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE ONLY -- CPU oracle for ``celerite.solver.CARMASolver``.

Restates cpp/include/celerite/carma.h in numpy / cmath complex arithmetic: the CARMA(p, q) model
in carma_pack's parameterisation, its Kalman-filter log-likelihood (Kelly et al. 2014, steps 2-12) and
the conversion to celerite coefficients.  Pure Python loops over n: for the sizes of the reference's own
test (tests/test_celerite.py:22-42, N = 100) up to a few thousand samples.

Parity status: the reference's C++ cannot be built here (Eigen absent), so this restatement is pinned
(tests/test_oracle.py) by
  * the identity the reference's own test asserts (test_celerite.py:22-42): the Kalman log-likelihood equals
    the celerite log-likelihood of ``get_celerite_coeffs()`` on that test's seeded inputs -- two
    independent computations (filter vs semiseparable Cholesky, the latter already pinned to the
    reference's goldens);
  * a dense multivariate-normal log-likelihood with K_ij = covariance(|t_i - t_j|) (carma.h:255-272), which
    uses neither the filter nor the coefficient conversion.
"""
import cmath
import math

import numpy as np


class CarmaInstability(RuntimeError):
    """celerite::carma_exception (exceptions.h:8-12)."""

    def __init__(self):
        RuntimeError.__init__(self, "CARMA model encountered an instability")


def roots_from_params(params):
    """carma.h:15-29: quadratic factors (log c, log b) -> root pairs; a trailing single -> -exp."""
    params = np.asarray(params, dtype=float)
    n = len(params)
    roots = np.zeros(n, dtype=complex)
    if n % 2 == 1:
        roots[n - 1] = -math.exp(params[n - 1])
    for i in range(0, n - 1, 2):
        b = complex(math.exp(params[i + 1]))
        c = complex(math.exp(params[i]))
        arg = cmath.sqrt(b * b - 4.0 * c)
        roots[i] = 0.5 * (-b + arg)
        roots[i + 1] = 0.5 * (-b - arg)
    return roots


def poly_from_roots(roots):
    """carma.h:31-44: coefficients (lowest power first) of prod (x - r_i)."""
    n = len(roots) + 1
    if n == 1:
        return np.ones(1, dtype=complex)
    poly = np.zeros(n, dtype=complex)
    poly[0] = -roots[0]
    poly[1] = 1.0
    for i in range(1, n - 1):
        for j in range(n - 1, 0, -1):
            poly[j] = poly[j - 1] - roots[i] * poly[j]
        poly[0] *= -roots[i]
    return poly


def _isclose(a, b):  # utils.h:16-20
    return abs(a - b) <= 1e-6


def _logsumexp(a, b):  # utils.h:22-25
    return b + cmath.log(1.0 + cmath.exp(a - b))


def _cpow(base, expo):
    """std::pow(std::complex<double>, double) as libstdc++ evaluates it: real power of a positive real,
    otherwise polar(exp(y log|x|), y arg x)."""
    if base.imag == 0.0 and base.real > 0.0:
        return complex(math.pow(base.real, expo))
    lg = cmath.log(base)
    return cmath.rect(math.exp(expo * lg.real), expo * lg.imag)


class CARMASolver(object):
    def __init__(self, log_sigma, arpars, mapars):  # carma.h:54-72
        arpars = np.atleast_1d(np.asarray(arpars, dtype=float))
        mapars = np.atleast_1d(np.asarray(mapars, dtype=float))
        self.sigma = math.exp(log_sigma)
        self.p, self.q = len(arpars), len(mapars)
        self.arroots = roots_from_params(arpars)
        self.maroots = roots_from_params(mapars)
        if self.q >= self.p:
            raise RuntimeError("dimension mismatch")
        self.lambda_base = np.array([cmath.exp(r) for r in self.arroots])
        self.alpha = poly_from_roots(self.arroots)
        self.beta = poly_from_roots(self.maroots)
        self.beta = self.beta / self.beta[0]
        self._setup()

    def _setup(self):  # carma.h:141-165
        p = self.p
        U = np.empty((p, p), dtype=complex)
        for i in range(p):
            for j in range(p):
                U[i, j] = self.arroots[j] ** i
        b = np.zeros(p, dtype=complex)
        b[:self.q + 1] = self.beta
        self.b = b.dot(U)
        e = np.zeros(p, dtype=complex)
        e[p - 1] = self.sigma
        Jv = np.linalg.solve(U, e)
        V = -np.outer(Jv, np.conj(Jv))
        for i in range(p):
            for j in range(p):
                V[i, j] /= self.arroots[i] + np.conj(self.arroots[j])
        self.V = V

    def get_celerite_coeffs(self):  # carma.h:74-139
        p, q = self.p, self.q
        ar, cr, a, b, c, d = [], [], [], [], [], []
        for k in range(p):
            rk = complex(self.arroots[k])
            term1 = cmath.log(self.beta[0])
            term2 = cmath.log(self.beta[0])
            for l in range(1, q + 1):
                term1 = _logsumexp(term1, cmath.log(self.beta[l]) + l * cmath.log(rk))
                term2 = _logsumexp(term2, cmath.log(self.beta[l]) + l * cmath.log(-rk))
            full = 2.0 * math.log(self.sigma) + term1 + term2 - cmath.log(complex(-rk.real))
            for l in range(p):
                if l != k:
                    rl = complex(self.arroots[l])
                    full -= cmath.log(rl - rk) + cmath.log(rl.conjugate() + rk)
            full = cmath.exp(full)
            if _isclose(full.imag, 0.0) and _isclose(rk.imag, 0.0):
                ar.append(0.5 * full.real)
                cr.append(-rk.real)
            else:
                conj = False
                for l in range(len(a)):
                    if (_isclose(a[l], full.real) and _isclose(b[l], -full.imag) and _isclose(c[l], -rk.real)
                            and _isclose(d[l], rk.imag)):
                        conj = True
                        break
                if not conj:
                    a.append(full.real)
                    b.append(full.imag)
                    c.append(-rk.real)
                    d.append(-rk.imag)
        return tuple(np.array(v, dtype=float) for v in (ar, cr, a, b, c, d))

    def log_likelihood(self, t, y, yerr):  # carma.h:221-239 with :167-219 inlined
        t, y, yerr = (np.asarray(v, dtype=float) for v in (t, y, yerr))
        n = len(t)
        if len(y) != n or len(yerr) != n:
            raise RuntimeError("dimension mismatch")
        p, b, V = self.p, self.b, self.V
        ll = n * math.log(2.0 * math.pi)
        x = np.zeros(p, dtype=complex)  # reset, :167-173
        P = V.copy()
        for i in range(n):
            # predict, :175-187
            expectation = 0.0
            variance = yerr[i] * yerr[i]
            for r in range(p):
                expectation += (b[r] * x[r]).real
                for s in range(p):
                    variance += (b[r] * P[r, s] * np.conj(b[s])).real
            if variance < 0.0:
                raise CarmaInstability()
            # update_state, :189-202
            K = np.zeros(p, dtype=complex)
            for r in range(p):
                for s in range(p):
                    K[r] += P[r, s] * np.conj(b[s]) / variance
                x[r] += (y[i] - expectation) * K[r]
            for r in range(p):
                for s in range(p):
                    P[r, s] -= variance * K[r] * np.conj(K[s])
            # advance_time, :204-219
            if i < n - 1:
                dt = t[i + 1] - t[i]
                lam = np.array([_cpow(complex(lb), dt) for lb in self.lambda_base])
                x = x * lam
                Pold = P
                P = V.copy()
                for r in range(p):
                    for s in range(p):
                        P[r, s] += lam[r] * (Pold[r, s] - V[r, s]) * np.conj(lam[s])
            resid = y[i] - expectation
            ll += resid * resid / variance + math.log(variance)
        return -0.5 * ll

    def covariance(self, tau):  # carma.h:255-272
        value = 0.0 + 0.0j
        for k in range(self.p):
            rk = complex(self.arroots[k])
            n1 = n2 = 0.0 + 0.0j
            for l in range(self.q + 1):
                n1 += self.beta[l] * rk ** l
                n2 += self.beta[l] * (-rk) ** l
            norm = n1 * n2 / rk.real
            for l in range(self.p):
                if l != k:
                    rl = complex(self.arroots[l])
                    norm /= (rl - rk) * (rl.conjugate() + rk)
            value += norm * cmath.exp(rk * tau)
        return -0.5 * self.sigma * self.sigma * value.real
