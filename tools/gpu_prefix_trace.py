"""Workload for rocprofv3: the headline shape with the multi-level prefix, one level of groups of argv[1] (default 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
coeffs, t, diag, y = make_inputs(1024, 100000, 2, 3, 42)
plan = batch.BatchedGP(1024, 100000, 2, 3)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
plan.set_prefix_mode("multilevel"); plan.set_prefix_plan(1, int(sys.argv[1]) if len(sys.argv) > 1 else 8)
for _ in range(6):
    plan.enqueue()
plan.results(); plan.close()
