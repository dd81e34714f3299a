# -*- coding: utf-8 -*-
"""Latency of CholeskySolver.dot (K z, one problem): GPU next to the CPU oracle, best of 3, parity."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref
print("# width | N | dot GPU ms | CPU ms | parity")
for JR, JC in [(2, 3), (2, 7), (0, 16)]:
    for N in (10000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        z = rng.randn(N)
        co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
              np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
        gen = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        yg, yc = s.dot(0.1, *co, *gen, t, z), r.dot(0.1, *co, *gen, t, z)
        tg, tc = best_of_3(lambda: s.dot(0.1, *co, *gen, t, z), 0.1), best_of_3(lambda: r.dot(0.1, *co, *gen, t, z), 0.1)
        print("width %2d  N=%6d  GPU %8.3f ms  CPU %8.3f ms  parity %.1e" % (JR + 2 * JC, N, tg * 1e3, tc * 1e3, np.max(np.abs(yg - yc)) / np.max(np.abs(yc))), flush=True)
