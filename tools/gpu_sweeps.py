# -*- coding: utf-8 -*-
"""Latency of the stored-factor consumers of the object API (one problem): dot_solve, solve, dot_L on the
GPU next to the CPU oracle, best of 3 (celerite/timer.py protocol), with parity."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref

print("# width | N | dot_solve GPU/CPU ms | solve GPU/CPU ms | dot_L GPU/CPU ms | parity (dot_solve, solve, dot_L)")
for JR, JC in [(2, 3), (2, 7), (4, 11), (0, 16)]:
    for N in (3000, 10000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        s.compute(*args); r.compute(*args)
        row = []
        par = []
        for name in ("dot_solve", "solve", "dot_L"):
            fg, fc = getattr(s, name), getattr(r, name)
            vg, vc = np.asarray(fg(y)).ravel(), np.asarray(fc(y)).ravel()
            par.append(np.max(np.abs(vg - vc)) / np.max(np.abs(vc)))
            row.append((best_of_3(lambda: fg(y), 0.1) * 1e3, best_of_3(lambda: fc(y), 0.1) * 1e3))
        print("width %2d  N=%6d  dot_solve %7.3f / %7.3f  solve %7.3f / %7.3f  dot_L %7.3f / %7.3f   parity %.1e %.1e %.1e"
              % (JR + 2 * JC, N, row[0][0], row[0][1], row[1][0], row[1][1], row[2][0], row[2][1], par[0], par[1], par[2]), flush=True)
