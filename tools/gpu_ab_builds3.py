"""Two builds on one box (see gpu_ab_builds.py): the materialising step at the headline shape (replay kernel ms, HBM fraction)."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)
plan = batch.BatchedGP(B, N, 2, 3)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
plan.enqueue(materialize=True); plan.synchronize()
fb = B * 8.0 * N * 25 + B * 24.0 * N
for _ in range(3):
    tot, k = plan.run_timed(5, materialize=True, relayout_each_step=False)
    print(os.environ["CLR_LIB"], "materialising step %.3f ms  replay %.3f ms = %.1f %% of 8 TB/s" % (tot / 5, k["replay"] / 5, fb / (k["replay"] / 5 * 1e-3) / 8e12 * 100), flush=True)
phi, u, W, D = plan.factor(7)
print("   factor checksum %.12e" % float(np.sum(W) + np.sum(D) + np.sum(phi) + np.sum(u)))
plan.close()
