# -*- coding: utf-8 -*-
"""Round 4: one series through CholeskySolver.compute + dot_solve at widths 16 / 32 by chunk count (CLR_SOLVER_WIDE_CHUNKS),
with the prefix as a parallel scan (wide_prefix_scan.hip) and as the sequential walk (CLR_WIDE_PREFIX_WALK=1)."""
import os, sys
os.environ["CLR_WIDE_SCAN_CAP"] = "1024"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import best_of_3
import celerite_amd
from oracle import ref
for JR, JC in [(2, 7), (0, 16)]:
    for N in (4096, 20000, 100000, 400000, 1000000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        r = ref.RefSolver()
        r.compute(*args); qc, lc = r.dot_solve(y), r.log_determinant()
        row = []
        for walk in (True, False):
            for chunks in ((None,) if walk else (None, 64, 128, 256, 384, 512, 768, 1024)):
                os.environ.pop("CLR_SOLVER_WIDE_CHUNKS", None); os.environ.pop("CLR_WIDE_PREFIX_WALK", None)
                if walk: os.environ["CLR_WIDE_PREFIX_WALK"] = "1"
                if chunks: os.environ["CLR_SOLVER_WIDE_CHUNKS"] = str(chunks)
                s = celerite_amd.CholeskySolver()
                def gpu_hinted():
                    s._hint_rhs(y); s.compute(*args); return s.dot_solve(y), s.log_determinant()
                qh, lh = gpu_hinted()
                th = best_of_3(gpu_hinted, 0.05)
                row.append("%s%s: %.2f ms (%.0e %.0e)" % ("walk " if walk else "", chunks or "rule", th * 1e3, abs(lh - lc) / abs(lc), abs(qh - qc) / abs(qc)))
        print("width %2d N=%6d  hinted compute + dot_solve:  %s" % (JR + 2 * JC, N, "  ".join(row)), flush=True)
os.environ.pop("CLR_SOLVER_WIDE_CHUNKS", None); os.environ.pop("CLR_WIDE_PREFIX_WALK", None)
