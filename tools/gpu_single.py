# -*- coding: utf-8 -*-
"""Latency of the reference's object API (one problem per call) across widths and lengths:
CholeskySolver.compute + dot_solve + log_determinant on the GPU next to the CPU oracle, best of 3
(celerite/timer.py protocol)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref

print("# width | N | GPU ms: compute + dot_solve | GPU ms: hint + compute + dot_solve (GP.log_likelihood) | CPU oracle ms | ratio | parity")
for JR, JC in [(1, 1), (2, 3), (2, 7), (4, 11), (0, 16)]:
    for N in (1000, 10000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        def gpu():
            s.compute(*args); return s.dot_solve(y), s.log_determinant()
        def gpu_hinted():   # what GP.log_likelihood does: the residual is announced before the factorisation
            s._hint_rhs(y); s.compute(*args); return s.dot_solve(y), s.log_determinant()
        def cpu():
            r.compute(*args); return r.dot_solve(y), r.log_determinant()
        (qg, lg), (qh, lh), (qc, lc) = gpu(), gpu_hinted(), cpu()
        tg, th, tc = best_of_3(gpu, 0.1), best_of_3(gpu_hinted, 0.1), best_of_3(cpu, 0.1)
        print("width %2d (%d real + %2d complex)  N=%6d  GPU %8.3f ms  GPU as GP.log_likelihood %8.3f ms  CPU %8.3f ms  CPU/GPU %5.2f  logdet %.1e  dot_solve %.1e / %.1e"
              % (JR + 2 * JC, JR, JC, N, tg * 1e3, th * 1e3, tc * 1e3, tc / th, abs(lg - lc) / abs(lc), abs(qg - qc) / abs(qc), abs(qh - qc) / abs(qc)), flush=True)
