# -*- coding: utf-8 -*-
"""Short single series at widths 16 / 32 through CholeskySolver: one sequential sweep (the rule below N = 2048) against a few chunks
with the parallel prefix (CLR_SOLVER_WIDE_CHUNKS)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref
for JR, JC in [(2, 7), (0, 16)]:
    for N in (700, 1000, 3000, 10000, 30000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        r = ref.RefSolver()
        def cpu():
            r.compute(*args); return r.dot_solve(y), r.log_determinant()
        qc, lc = cpu()
        tc = best_of_3(cpu, 0.05)
        row = []
        cap = 1024 if JR + 2 * JC <= 16 else 512
        for chunks in (None,) + tuple(sorted(set(max(8, min(cap, N // L)) for L in (256, 160, 128, 96, 64, 48, 32)))):
            os.environ.pop("CLR_SOLVER_WIDE_CHUNKS", None)
            if chunks: os.environ["CLR_SOLVER_WIDE_CHUNKS"] = str(chunks)
            s = celerite_amd.CholeskySolver()
            def gpu_hinted():
                s._hint_rhs(y); s.compute(*args); return s.dot_solve(y), s.log_determinant()
            qh, lh = gpu_hinted()
            th = best_of_3(gpu_hinted, 0.05)
            row.append("%s: %.3f (%.0e)" % (chunks or "rule", th * 1e3, max(abs(lh - lc) / abs(lc), abs(qh - qc) / abs(qc))))
        print("width %2d N=%5d  CPU %.3f ms | GPU ms by chunks  %s" % (JR + 2 * JC, N, tc * 1e3, "  ".join(row)), flush=True)
os.environ.pop("CLR_SOLVER_WIDE_CHUNKS", None)
