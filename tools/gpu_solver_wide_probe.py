# -*- coding: utf-8 -*-
"""CholeskySolver.compute on one series of 1e5 samples at (J_real, J_comp) shapes of padded width 32: time by chunk count and prefix form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
for (JR, JC) in ((4, 11), (2, 7)):
    N = 100000
    rng = np.random.RandomState(JR * 100 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    yerr = rng.uniform(0.3, 0.5, N)
    y = rng.randn(N)
    args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
            np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
            np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
    row = []
    for walk in (False,):
        for chunks in (128, 300, 350, 390, 391, 512):
            os.environ.pop("CLR_WIDE_PREFIX_WALK", None)
            if walk: os.environ["CLR_WIDE_PREFIX_WALK"] = "1"
            os.environ["CLR_SOLVER_WIDE_CHUNKS"] = str(chunks)
            s = celerite_amd.CholeskySolver()
            s.compute(*args)
            t0 = time.perf_counter(); s.compute(*args); dt = time.perf_counter() - t0
            row.append("%s%d: %.2f ms (logdet %.12e)" % ("walk " if walk else "", chunks, dt * 1e3, s.log_determinant()))
    print("(%d,%d): %s" % (JR, JC, "  ".join(row)), flush=True)
