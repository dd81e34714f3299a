"""Workload for rocprofv3: three gradients of the headline batch (B = 1024, N = 1e5, width 8, 17 partials) on a
resident plan (clr_batch_grad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
coeffs, t, diag, y = make_inputs(1024, 100000, 2, 3, 42)
plan = batch.BatchedGP(1024, 100000, 2, 3)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
for _ in range(3):
    v, g, st = plan.grad_log_likelihood()
print("ok", int((st == 0).sum()), "fallbacks", plan.grad_fallbacks())
plan.close()
