# -*- coding: utf-8 -*-
"""Workload for rocprofv3: the stored-factor operations of one solver object at N = 1e5 (width from argv): dot_solve, solve, dot_L, dot, predict."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
JR, JC = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 3)
N = 100000
rng = np.random.RandomState(JR * 100 + JC)
t = np.sort(rng.uniform(0, 0.05 * N, N))
yerr = rng.uniform(0.3, 0.5, N)
y = rng.randn(N)
co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
      np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
e, e2 = np.empty(0), np.empty((0, 0))
args = (0.0,) + co + (e, e2, e2, t, yerr ** 2)
s = celerite_amd.CholeskySolver()
s.compute(*args)
xs = np.sort(rng.uniform(t[0], t[-1], 20000))
ops = {"dot_solve": lambda: s.dot_solve(y), "solve": lambda: s.solve(y), "dot_L": lambda: s.dot_L(y),
       "dot": lambda: s.dot(0.0, *co, e, e2, e2, t, y[:, None]), "predict": lambda: s.predict(y, xs)}
for name, f in ops.items():
    f()
    t0 = time.perf_counter()
    for _ in range(5): f()
    print("width %d %-9s %.3f ms" % (JR + 2 * JC, name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
