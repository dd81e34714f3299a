# -*- coding: utf-8 -*-
"""The reference's OWN benchmark grid (examples/benchmark/run.py:37-38, 66-84, 131-138: J = 2^k terms -> width 2 J, one
real term + complex terms (0.1, 2.0, 1.6); t = sort(rand(N)), yerr ~ U(0.1, 0.2), y = sin t; `compute` and
`log_likelihood` timed separately with its timer's best-of-3) through this build's object API (celerite_amd.GP) and, beside
it, the CPU oracle's compute + dot_solve on the same inputs (one core).  Not a BASELINE config: the single-problem
latencies behind the batched numbers."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd
from celerite_amd import terms
from oracle import ref


def best_of_3(fn, min_time=0.05):
    best = np.inf
    for _ in range(3):
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= min_time:
                break
        best = min(best, dt / n)
    return best


np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX))
yerr_all = np.random.uniform(0.1, 0.2, NMAX)
y_all = np.sin(t_all)
E, E2 = np.empty(0), np.empty((0, 0))
print("# width = 2 J; ms per call; cpu = oracle/celerite_ref.c on one core (compute | dot_solve)")
print("%5s %8s | %10s %10s | %10s %10s | %8s %8s" % ("width", "N", "gpu comp", "gpu ll", "cpu comp", "cpu ll", "comp x", "ll x"))
for jpow in range(0, 9):
    j = 2 ** jpow
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2):
        kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2):
        kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    coeffs = kernel.coefficients
    width = len(coeffs[0]) + 2 * len(coeffs[2])
    assert width == 2 * j
    gp = celerite_amd.GP(kernel)
    for npow in (10, 13, 16, 19):
        N = 2 ** npow
        if width * width * N > 2 ** 34:      # (keeps the CPU leg and the sequential wide kernels within seconds)
            continue
        t, yerr, y = t_all[:N], yerr_all[:N], y_all[:N]
        gp.compute(t, yerr)
        ll = gp.log_likelihood(y)
        g_comp = best_of_3(lambda: gp.compute(t, yerr))
        g_ll = best_of_3(lambda: gp.log_likelihood(y))
        r = ref.RefSolver()
        cs = [np.asarray(c, dtype=float) for c in coeffs]
        d = yerr ** 2
        c_comp = best_of_3(lambda: r.compute(0.0, *cs, E, E2, E2, t, d))
        c_ll = best_of_3(lambda: r.dot_solve(y))
        ll0 = -0.5 * (r.dot_solve(y) + r.log_determinant() + N * np.log(2 * np.pi))
        print("%5d %8d | %10.3f %10.3f | %10.3f %10.3f | %8.2f %8.2f   rel %.1e" % (
            width, N, g_comp * 1e3, g_ll * 1e3, c_comp * 1e3, c_ll * 1e3, c_comp / g_comp, c_ll / g_ll, abs(ll - ll0) / abs(ll0)), flush=True)
