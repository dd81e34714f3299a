"""Finer chunk sweep of the plan gradient around the wave counts that fill the SIMDs exactly (two directions per wave)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from celerite_amd import batch
from _cases import synthetic, coeffs_of
for (N, JR, JC, sweep) in ((100000, 0, 8, (48, 56, 58, 60, 62, 64, 90, 112, 116, 120, 124)), (100000, 0, 16, (24, 28, 30, 31, 32, 46, 48, 60, 62)),
                           (100000, 2, 5, (48, 60, 64, 78, 90, 120, 128, 150)), (100000, 0, 6, (64, 78, 128, 157))):
    case = synthetic(1, N, JR, JC, "bench", seed=JR + JC)
    plan = batch.BatchedGP(1, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case), jitter=0.01)
    row = []
    for nchunk in sweep:
        plan.set_chunks(nchunk)
        plan.set_coefficients(*coeffs_of(case), jitter=0.01)
        plan.grad_log_likelihood()
        t0 = time.perf_counter()
        for _ in range(4):
            plan.grad_log_likelihood()
        row.append("%d: %.2f" % (plan.chunks[0], (time.perf_counter() - t0) / 4 * 1e3))
    G = 1 + 2 * JR + 4 * JC
    print("N=%d (%d,%d) G=%d waves per chunk %d; ms per gradient by chunk count: %s" % (N, JR, JC, G, (G + 1) // 2, "  ".join(row)), flush=True)
    plan.close()
