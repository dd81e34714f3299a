# -*- coding: utf-8 -*-
"""dot_solve / solve of one solver object (stored factor, wave-per-chunk sweeps) by chunk count (CLR_WSWEEP_CHUNKS) and run length of the two-level
prefix (CLR_WSWEEP_RUN; 0 = one walk over all chunks), against the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from oracle import ref
for (JR, JC) in ((2, 3), (2, 7), (0, 16)):
    for N in (5000, 20000, 100000, 400000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
              np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
        e, e2 = np.empty(0), np.empty((0, 0))
        args = (0.0,) + co + (e, e2, e2, t, yerr ** 2)
        r = ref.RefSolver(); r.compute(*args); x0 = r.solve(y); q0 = r.dot_solve(y)
        s = celerite_amd.CholeskySolver(); s.compute(*args)
        row = []
        for chunks, run in ((None, 0), (512, 8), (768, 8), (768, 12), (768, 16), (1024, 8), (1024, 12), (1024, 16), (1536, 8), (1536, 12), (1536, 16)):
            os.environ.pop("CLR_WSWEEP_CHUNKS", None); os.environ.pop("CLR_WSWEEP_RUN", None)
            if chunks: os.environ["CLR_WSWEEP_CHUNKS"] = str(chunks)
            if run is not None: os.environ["CLR_WSWEEP_RUN"] = str(run)
            x = s.solve(y); q = s.dot_solve(y)
            t0 = time.perf_counter()
            for _ in range(5): q = s.dot_solve(y)
            d1 = (time.perf_counter() - t0) / 5
            t0 = time.perf_counter()
            for _ in range(5): x = s.solve(y)
            d2 = (time.perf_counter() - t0) / 5
            row.append("%s/%s: %.3f %.3f (%.0e)" % (chunks or "rule", "rule" if run is None else run, d1 * 1e3, d2 * 1e3,
                                                    max(abs(q - q0) / abs(q0), np.max(np.abs(x.ravel() - x0.ravel())) / np.max(np.abs(x0)))))
        print("width %2d N=%6d dot_solve solve ms by chunks/run: %s" % (JR + 2 * JC, N, "  ".join(row)), flush=True)
os.environ.pop("CLR_WSWEEP_CHUNKS", None); os.environ.pop("CLR_WSWEEP_RUN", None)
