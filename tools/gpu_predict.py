# -*- coding: utf-8 -*-
"""Latency of CholeskySolver.predict (one problem) across widths: GPU next to the CPU oracle, best of 3, parity."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref

print("# width | N | M | predict GPU ms | CPU ms | parity")
for JR, JC in [(2, 3), (2, 7), (0, 16)]:
    for N, M in ((10000, 3000), (100000, 20000)):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        xs = np.linspace(t[0] - 1, t[-1] + 1, M)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        s.compute(*args); r.compute(*args)
        pg, pc = s.predict(y, xs), r.predict(y, xs)
        tg, tc = best_of_3(lambda: s.predict(y, xs), 0.1), best_of_3(lambda: r.predict(y, xs), 0.1)
        print("width %2d  N=%6d  M=%5d  GPU %8.3f ms  CPU %8.3f ms  parity %.1e" % (JR + 2 * JC, N, M, tg * 1e3, tc * 1e3, np.max(np.abs(pg - pc)) / max(1.0, np.max(np.abs(pc)))), flush=True)
