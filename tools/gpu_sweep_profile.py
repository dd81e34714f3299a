# -*- coding: utf-8 -*-
"""Workload for rocprofv3: the stored-factor sweeps of one solver object (width from argv, N = 1e5)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
JR, JC = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 16)
N = 100000
rng = np.random.RandomState(JR * 100 + JC)
t = np.sort(rng.uniform(0, 0.05 * N, N))
yerr = rng.uniform(0.3, 0.5, N)
y = rng.randn(N)
args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
        np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
        np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
s = celerite_amd.CholeskySolver()
s.compute(*args)
for _ in range(20):
    s.dot_solve(y); s.solve(y); s.dot_L(y)
