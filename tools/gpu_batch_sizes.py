# -*- coding: utf-8 -*-
"""Throughput of the headline shape (N = 1e5, width 8) against the batch size on one GPU (device-only steps)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
print("# B | chunks | ms per step | loglik/s | HBM in use GiB")
for B in (64, 256, 512, 1024, 2048, 4096, 8192):
    coeffs, t, diag, y = make_inputs(B, 100000, 2, 3, seed=B)
    plan = batch.BatchedGP(B, 100000, 2, 3)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(5, relayout_each_step=False)
    free, total = batch.device_memory()
    print("%5d  %s  %8.3f ms  %9.0f /s  %6.1f  %s" % (B, plan.chunks, tot / 5, B / (tot / 5 * 1e-3), (total - free) / 2**30, plan.summarize_kernel()), flush=True)
    plan.close()
    del coeffs, t, diag, y
