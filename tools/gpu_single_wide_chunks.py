# -*- coding: utf-8 -*-
"""Chunk-count sweep of the single-solver wide route (CLR_EXPERIMENT_CHUNKS was a temporary knob in
clr_solver_compute; the rule derived from profiles/r02y_single_wide_chunks.txt is now built in)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref
for JR, JC in [(2, 7), (4, 11), (0, 16)]:
    for N in (10000, 30000, 100000):
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
        def gpu_hinted():
            s._hint_rhs(y); s.compute(*args); return s.dot_solve(y), s.log_determinant()
        r.compute(*args); qc, lc = r.dot_solve(y), r.log_determinant()
        (qg, lg), (qh, lh) = gpu(), gpu_hinted()
        tg, th = best_of_3(gpu, 0.1), best_of_3(gpu_hinted, 0.1)
        print("chunks<=%s width %2d N=%6d  compute+dot_solve %7.3f ms  hinted %7.3f ms  logdet %.1e quad %.1e %.1e" % (os.environ.get("CLR_EXPERIMENT_CHUNKS", "16"), JR + 2 * JC, N, tg * 1e3, th * 1e3, abs(lg - lc) / abs(lc), abs(qg - qc) / abs(qc), abs(qh - qc) / abs(qc)), flush=True)
