# -*- coding: utf-8 -*-
"""CholeskySolver.predict (N samples, M = 2e4 sorted prediction points) by chunk count of its two scans (CLR_PREDICT_CHUNKS), against the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from oracle import ref
for (JR, JC) in ((2, 3), (2, 7), (0, 16)):
    for N in (10000, 100000):
        rng = np.random.RandomState(JR * 100 + JC)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
              np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
        e, e2 = np.empty(0), np.empty((0, 0))
        args = (0.0,) + co + (e, e2, e2, t, yerr ** 2)
        xs = np.sort(rng.uniform(t[0] - 1.0, t[-1] + 1.0, 20000))
        r = ref.RefSolver(); r.compute(*args); p0 = r.predict(y, xs)
        s = celerite_amd.CholeskySolver(); s.compute(*args)
        row = []
        for chunks in (None, 576, 1024, 2048, 4096, 8192):
            os.environ.pop("CLR_PREDICT_CHUNKS", None)
            if chunks: os.environ["CLR_PREDICT_CHUNKS"] = str(chunks)
            p = s.predict(y, xs)
            t0 = time.perf_counter()
            for _ in range(5): p = s.predict(y, xs)
            dt = (time.perf_counter() - t0) / 5
            row.append("%s: %.3f ms (%.0e)" % (chunks or "rule", dt * 1e3, np.max(np.abs(p - p0)) / np.max(np.abs(p0))))
        print("width %2d N=%6d predict: %s" % (JR + 2 * JC, N, "  ".join(row)), flush=True)
os.environ.pop("CLR_PREDICT_CHUNKS", None)
