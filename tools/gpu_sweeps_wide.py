# -*- coding: utf-8 -*-
"""Latency of the stored-factor sweeps of the object API at widths 8..64, N = 1e5 (and short series), next to the CPU
oracle: dot_solve, solve, dot_L.  Round 3: widths 33..64 and N >= 512 above width 8 run as chunked scans."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref

print("# width | N | dot_solve GPU / CPU ms | solve GPU / CPU ms | dot_L GPU / CPU ms | parity (dot_solve rel, solve rel)")
for JR, JC, N in [(2, 3, 100000), (0, 16, 100000), (0, 20, 100000), (0, 24, 100000), (0, 32, 100000),
                  (4, 11, 1000), (0, 32, 1000), (0, 32, 10000)]:
    rng = np.random.RandomState(JR * 100 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
          np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
    gen = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
    s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
    s.compute(0.0, *co, *gen, t, diag); r.compute(0.0, *co, *gen, t, diag)
    b = rng.randn(N)
    q, q0 = s.dot_solve(b), r.dot_solve(b)
    x, x0 = s.solve(b), r.solve(b)
    row = []
    for f, g in ((lambda: s.dot_solve(b), lambda: r.dot_solve(b)), (lambda: s.solve(b), lambda: r.solve(b)),
                 (lambda: s.dot_L(b), lambda: r.dot_L(b))):
        row += [best_of_3(f, 0.05) * 1e3, best_of_3(g, 0.05) * 1e3]
    print("width %2d  N=%6d  dot_solve %7.3f / %7.3f   solve %7.3f / %7.3f   dot_L %7.3f / %7.3f   parity %.1e %.1e"
          % (JR + 2 * JC, N, *row, abs(q - q0) / abs(q0), np.max(np.abs(x - x0)) / np.max(np.abs(x0))), flush=True)
