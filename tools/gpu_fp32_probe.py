"""fp32-state sweep against the fp64 sweep at BASELINE config 5's shape (width 32): error and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs
from celerite_amd import batch
for B, N in [(256, 2000), (256, 20000), (256, 100000), (2048, 100000)]:
    coeffs, t, diag, y = make_inputs(B, N, 0, 16, 11, d_spread=True)
    plan = batch.BatchedGP(B, N, 0, 16)
    plan.set_chunks(1)           # the plain sequential sweep, one wave per problem (fp64 reference of the comparison)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(2)
    ll, ld, q, st = plan.results()
    ld32, q32, ms = plan.fp32_probe()
    plan.close()
    print("B=%4d N=%6d width 32: fp64 sweep %8.2f ms  fp32-state sweep %8.2f ms (%.2fx)  |  fp32 vs fp64: log det max %.2e median %.2e, quadratic form max %.2e median %.2e"
          % (B, N, tot / 2, ms, tot / 2 / ms, np.max(np.abs(ld32 - ld) / np.abs(ld)), np.median(np.abs(ld32 - ld) / np.abs(ld)),
             np.max(np.abs(q32 - q) / np.abs(q)), np.median(np.abs(q32 - q) / np.abs(q))), flush=True)
