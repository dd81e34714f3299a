# -*- coding: utf-8 -*-
"""Small batches of long series through the batch API (N = 1e5): ms per evaluation (real loop: coefficients in, results out) at widths 8 and 16."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
for (JR, JC) in ((2, 3), (0, 8), (0, 16)):
    row = []
    for B in (1, 2, 4, 8, 16, 32, 64, 128):
        coeffs, t, diag, y = make_inputs(B, 100000, JR, JC, seed=B, d_spread=(JC >= 8)) if JC >= 8 else make_inputs(B, 100000, JR, JC, seed=B)
        plan = batch.BatchedGP(B, 100000, JR, JC)
        plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
        plan.log_likelihood()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.set_coefficients(*coeffs); ll, ld, q, st = plan.log_likelihood()
        dt = (time.perf_counter() - t0) / 5
        row.append("B=%d %s: %.3f ms (routes %s)" % (B, plan.chunks, dt * 1e3, np.bincount(plan.exact_levels(), minlength=3).tolist()))
        plan.close()
    print("width %d: %s" % (JR + 2 * JC, "  ".join(row)), flush=True)
