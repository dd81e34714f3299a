# -*- coding: utf-8 -*-
"""Workload for rocprofv3 --hip-trace --kernel-trace --stats: BASELINE config 0 (N = 1000, width 3) through the object
API, 300 GP.log_likelihood-style evaluations (hint + compute + dot_solve + log_determinant)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from celerite_amd import terms
N = 1000
rng = np.random.RandomState(3)
t = np.sort(rng.uniform(0, 10, N)); yerr = rng.uniform(0.1, 0.2, N); y = np.sin(t)
k = terms.RealTerm(0.1, 0.5) + terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
co = k.coefficients
e = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
s = celerite_amd.CholeskySolver()
def once():
    s._hint_rhs(y); s.compute(0.0, *co, *e, t, yerr ** 2); return s.dot_solve(y) + s.log_determinant()
for _ in range(20): once()
t0 = time.perf_counter()
for _ in range(300): once()
print("ms per evaluation: %.4f" % ((time.perf_counter() - t0) / 300 * 1e3))
