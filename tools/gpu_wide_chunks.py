# -*- coding: utf-8 -*-
"""Development probe: chunk-count sweep of the wide scan (width 32, N = 1e5)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite_amd import batch  # noqa: E402

rng = np.random.RandomState(3)
N, JR, JC = 100000, 0, 16
for B in (256, 1024, 2048):
    t = np.sort(rng.rand(B, N), axis=1); sig = rng.uniform(0.1, 0.2, (B, N)); y = np.sin(t)
    ac = np.exp(0.1 + 0.1 * rng.randn(B, JC)); bc = np.zeros((B, JC))
    cc = np.exp(2.0 + 0.1 * rng.randn(B, JC)); dc = np.exp(rng.uniform(0.0, 3.0, (B, JC)))
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, sig ** 2, y)
    plan.set_coefficients(np.empty((B, 0)), np.empty((B, 0)), ac, bc, cc, dc)
    for nch in (1, 2, 3, 4, 6, 8, 16):
        plan.set_chunks(nch)
        plan.log_likelihood()
        tot, k = plan.run_timed(2)
        print("B=%d width 32 chunks %2d: %.2f ms  (%s)" % (B, plan.chunks[0], tot / 2, " ".join("%s %.2f" % (a, v / 2) for a, v in k.items() if v > 0.01)), flush=True)
    plan.close()
