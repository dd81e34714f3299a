# -*- coding: utf-8 -*-
"""Latency of CholeskySolver.grad_log_likelihood (one problem, forward-mode tangents, one wave per partial)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
print("# width | N | partials | grad_log_likelihood GPU ms | compute + dot_solve GPU ms")
for JR, JC in [(1, 1), (2, 3), (2, 7)]:
    for N in (1000, 10000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N)); yerr = rng.uniform(0.3, 0.5, N); y = rng.randn(N)
        co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
              np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
        e = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
        s = celerite_amd.CholeskySolver()
        s.grad_log_likelihood(0.1, *co, *e, t, y, yerr ** 2)
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps): v, g = s.grad_log_likelihood(0.1, *co, *e, t, y, yerr ** 2)
        tg = (time.perf_counter() - t0) / reps
        s.compute(0.1, *co, *e, t, yerr ** 2)
        t0 = time.perf_counter()
        for _ in range(reps): s.compute(0.1, *co, *e, t, yerr ** 2); s.dot_solve(y)
        tc = (time.perf_counter() - t0) / reps
        print("width %2d  N=%6d  partials %2d  grad %9.3f ms  value-only %8.3f ms" % (JR + 2 * JC, N, len(g), tg * 1e3, tc * 1e3), flush=True)
