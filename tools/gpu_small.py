# -*- coding: utf-8 -*-
"""Development probe: launch-bound regime -- small batches evaluated back to back."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from celerite_amd import batch  # noqa: E402
from _cases import synthetic, coeffs_of  # noqa: E402

for (B, N, JR, JC) in [(16, 500, 1, 1), (64, 1000, 1, 1), (256, 10000, 2, 1), (1024, 1000, 2, 3), (8, 100000, 2, 3)]:
    case = synthetic(B, N, JR, JC, "bench", seed=1)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    co = coeffs_of(case)
    plan.set_coefficients(*co)
    plan.log_likelihood()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.enqueue()
    plan.synchronize()
    t_enq = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.set_coefficients(*co)
        plan.enqueue()
        plan.results()
    t_full = (time.perf_counter() - t0) / reps
    print("B=%4d N=%6d width %d chunks %s: enqueue-only %.1f us per evaluation (%.0f loglik/s) ; set_coefficients + enqueue + results %.1f us" % (
        B, N, JR + 2 * JC, plan.chunks, t_enq * 1e6, B / t_enq, t_full * 1e6), flush=True)
    plan.close()
