"""The accuracy family (paper/figures/error/error.py:24-25) at the headline shape for the profiler: a few
evaluations of the warm-started recurrence (B = 1024, N = 1e5, width 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_accuracy
from celerite_amd import batch
coeffs, t, diag, y = make_inputs_accuracy(1024, 100000, 2, 3, 4242)
plan = batch.BatchedGP(1024, 100000, 2, 3)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
for _ in range(6):
    plan.enqueue()
ll, ld, q, st = plan.results()
print("warm", plan.warm_start(), "status ok", int((st == 0).sum()), "ll[0]", ll[0])
plan.close()
