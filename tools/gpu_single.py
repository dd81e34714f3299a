# -*- coding: utf-8 -*-
"""Development probe: latency of the single-solver API (the reference's own object API)
at the sizes of BASELINE config 1 and of the large-N benchmark rows."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd  # noqa: E402
from celerite_amd import GP, terms  # noqa: E402
from oracle import ref  # noqa: E402


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


rng = np.random.RandomState(1)
for N in (1000, 10000, 100000, 1000000):
    t = np.sort(rng.rand(N)); yerr = rng.uniform(0.1, 0.2, N); y = np.sin(t)
    kernel = terms.RealTerm(1.0, 0.1) + terms.RealTerm(1.1, 0.2)
    for _ in range(3):
        kernel += terms.ComplexTerm(log_a=0.1, log_c=2.0, log_d=1.6 + 0.1 * rng.randn())
    gp = GP(kernel)
    co = kernel.coefficients
    s = celerite_amd.CholeskySolver()
    gen = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
    tc = best(lambda: s.compute(0.0, *co, *gen, t, yerr ** 2))
    td = best(lambda: s.dot_solve(y))
    ts_ = best(lambda: s.solve(y))
    tl = best(lambda: s.dot_L(y))
    tp = best(lambda: s.predict(y, np.linspace(0, 1, 500)))
    tg = best(lambda: s.grad_log_likelihood(0.0, *co, *gen, t, y, yerr ** 2), reps=2) if N <= 100000 else float("nan")
    gp.compute(t, yerr)
    tll = best(lambda: (gp.compute(t, yerr), gp.log_likelihood(y)))
    r = ref.RefSolver()
    t0 = time.perf_counter(); r.compute(0.0, *co, *gen, t, yerr ** 2); q0 = r.dot_solve(y); tcpu = time.perf_counter() - t0
    print("N=%7d width 8: compute %.2f ms dot_solve %.2f solve %.2f dot_L %.2f predict(500) %.2f grad %.1f | GP compute+loglike %.2f ms ; "
          "CPU oracle compute+dot_solve %.2f ms ; rel err logdet %.1e quad %.1e" % (
              N, tc * 1e3, td * 1e3, ts_ * 1e3, tl * 1e3, tp * 1e3, tg * 1e3, tll * 1e3, tcpu * 1e3,
              abs(s.log_determinant() - r.log_determinant()) / abs(r.log_determinant()), abs(s.dot_solve(y) - q0) / abs(q0)), flush=True)
